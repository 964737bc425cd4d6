"""Device-resident corpus (nnr_amd.corpus / csrc/corpus.hip, through the C-ABI) against the goldens captured from the
reference's own MIND_corpus.py / MIND_dataset.py and against the numpy oracle.  Byte / integer / index work: bit-exact;
the fp32 graph normalisation too (correctly rounded div, sqrt, mul)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TAGS = ['tiny_h50_sym', 'tiny_h8_asym', 'tiny_h8_none', 'tiny_h8_noself']


def load(tag):
    return dict(np.load(os.path.join(GOLD, 'corpus_%s.npz' % tag)))


def _check_batch(got, exp_list):
    assert len(got) == 21
    for k, (g, e) in enumerate(zip(got, exp_list)):
        g = g.cpu().numpy()
        assert g.dtype == e.dtype and g.shape == e.shape, (k, g.dtype, e.dtype, g.shape, e.shape)
        np.testing.assert_array_equal(g, e, err_msg='batch field %d' % k)


@pytest.mark.parametrize('tag', TAGS)
@pytest.mark.parametrize('graph', ['build', 'table'])
def test_train_batch_equals_reference_dataloader(tag, graph):
    from nnr_amd.corpus import DeviceCorpus
    c = load(tag)
    dc = DeviceCorpus(c, 'cuda', int(c['category_num']), graph=graph, norm=str(c['norm']))
    dc.set_samples(c['train_samples'])
    got = dc.train_batch(c['batch_index'])
    _check_batch(got, [c['batch_%02d' % k] for k in range(21)])
    # every behaviour of the fixture, in one batch, against the oracle
    from oracle import corpus_oracle as CO
    idx = np.arange(c['beh_user'].shape[0], dtype=np.int32)
    _check_batch(dc.train_batch(idx), CO.train_batch(c, idx))


@pytest.mark.parametrize('tag', TAGS)
def test_history_graph_equals_reference_preprocess(tag):
    """The graph builder on the RAW history of every behaviours.tsv line (the reference keeps the last H items)."""
    from nnr_amd.corpus import history_graph
    c = load(tag)
    H, K, norm = int(c['max_history_num']), int(c['category_num']), str(c['norm'])
    lens, raw = c['line_history_len'], c['line_history_category']
    cats = np.zeros((lens.shape[0], H), dtype=np.int32)
    hmask = np.zeros((lens.shape[0], H), dtype=bool)
    for i, n in enumerate(lens):
        off, m = max(0, n - H), min(n, H)                      # MIND_corpus.py:188-189, 345-346
        cats[i, :m] = raw[i, off:off + m]
        hmask[i, :m] = True
    g, cm, ci = history_graph(torch.from_numpy(cats).cuda(), torch.from_numpy(hmask).cuda(), K, norm)
    np.testing.assert_array_equal(g.cpu().numpy(), c['train_user_history_graph'])
    np.testing.assert_array_equal(cm.cpu().numpy(), c['train_user_history_category_mask'])
    np.testing.assert_array_equal(ci.cpu().numpy(), c['train_user_history_category_indices'])


@pytest.mark.parametrize('norm', ['symmetric', 'asymmetric', 'none', 'none_noself'])
def test_history_graph_mind_shape_random_vs_oracle(norm):
    """H = 50, 18 categories (G = 68), 512 random histories incl. empty / single / full ones, against the numpy restatement."""
    from nnr_amd.corpus import history_graph
    from oracle import corpus_oracle as CO
    rng = np.random.default_rng(5)
    B, H, K = 512, 50, 18
    counts = rng.integers(0, H + 1, size=B)
    counts[:4] = [0, 1, H, H]
    cats = rng.integers(0, K, size=(B, H)).astype(np.int32)
    cats[3] = 7                                             # one cluster holding all 50
    hmask = np.arange(H)[None, :] < counts[:, None]
    cats[~hmask] = 0                                        # padded slots carry the PAD news' category
    g, cm, ci = history_graph(torch.from_numpy(cats).cuda(), torch.from_numpy(hmask).cuda(), K, norm)
    g, cm, ci = g.cpu().numpy(), cm.cpu().numpy(), ci.cpu().numpy()
    for b in range(B):
        eg, em, ei = CO.history_graph(cats[b], int(counts[b]), H, K, norm)
        np.testing.assert_array_equal(g[b], eg, err_msg='graph %d' % b)
        np.testing.assert_array_equal(cm[b], em)
        np.testing.assert_array_equal(ci[b], ei)
    # properties that hold at any size: symmetric normalisation keeps symmetry; asymmetric rows sum to 1
    if norm == 'symmetric':
        assert np.array_equal(g, np.transpose(g, (0, 2, 1)))
    if norm == 'asymmetric':
        nz = counts > 0
        np.testing.assert_allclose(g[nz].sum(axis=2), 1.0, atol=1e-6)


def test_full_size_batch_roundtrip_properties_and_model_step():
    """BASELINE sizes (batch 64, H = 50, T = 32, C = 128, 20 000 news): gather == numpy fancy indexing; graph built on the
    device == the pre-built table; the model's logits are identical whichever way the batch was produced."""
    from nnr_amd.config import make_config
    from nnr_amd.corpus import from_synth
    from nnr_amd.model import Model
    from nnr_amd.synth import SynthSpec, SynthCorpus
    spec = SynthSpec(vocabulary_size=60000)
    synth = SynthCorpus(spec)
    rng = np.random.default_rng(9)
    dc = from_synth(synth, 256, np.random.default_rng(9), 'cuda', graph='build')
    dt = from_synth(synth, 256, np.random.default_rng(9), 'cuda', graph='table')
    idx = rng.permutation(256)[:64].astype(np.int32)
    a, b = dc.train_batch(idx), dt.train_batch(idx)
    for k, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), 'field %d: device-built graph batch != table batch' % k
    hist = dc.t['beh_history'].cpu().numpy()[idx]
    np.testing.assert_array_equal(a[6].cpu().numpy(), synth.content_text[hist])
    np.testing.assert_array_equal(a[4].cpu().numpy(), synth.title_mask[hist])
    samp = dc.samples.cpu().numpy()[idx]
    np.testing.assert_array_equal(a[18].cpu().numpy(), synth.content_text[samp])
    cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=64'], corpus_sizes=dict(vocabulary_size=60000))
    torch.manual_seed(0)
    model = Model(cfg, torch.randn(cfg.vocabulary_size, cfg.word_embedding_dim) * 0.3)
    model.initialize()
    model = model.cuda().eval()
    with torch.no_grad():
        la = model(*[t.clone() for t in a])
        lb = model(*[t.clone() for t in b])
    assert torch.equal(la, lb) and bool(torch.isfinite(la).all())


def test_negative_sampling_restatement():
    from nnr_amd.corpus import negative_sampling
    c = load('tiny_h50_sym')
    # the fixture's samples came from the reference's sampler drawing from RandomState(3).randint; replay it
    # (click, negatives) per behaviour are not in the fixture, so check the structural rules on synthetic lists instead
    rs = np.random.RandomState(3)
    beh = [(5, [7]), (6, [8, 9]), (1, [2, 3, 4, 10]), (11, list(range(20, 40)))]
    s = negative_sampling(beh, 4, rs.randint)
    assert s[0].tolist() == [5, 7, 7, 7, 7] and s[1].tolist() == [6, 8, 9, 8, 9] and s[2].tolist() == [1, 2, 3, 4, 10]
    assert s[3, 0] == 11 and len(set(s[3, 1:].tolist())) == 4 and all(20 <= v < 40 for v in s[3, 1:])
    assert c['train_samples'].shape[1] == 5


def test_training_epoch_over_device_corpus_matches_oracle_loop():
    """End to end over the reference-built corpus (tiny MIND tree): DistributedSampler-order batches produced by DeviceCorpus
    (graphs built on the device) -> Trainer.train_step, against the oracle's train_step on numpy-gathered batches of the same
    behaviours -- the loop of trainer.py:78-120 with both of its ends replaced.  Dropout off; loss per step within 5e-5."""
    from nnr_amd import dp
    from nnr_amd.corpus import DeviceCorpus
    from nnr_amd.model import Model
    from nnr_amd.trainer import Trainer
    from oracle import corpus_oracle as CO, nnr_oracle as O
    c = load('tiny_h50_sym')
    V = int(max(c['news_title_text'].max(), c['news_abstract_text'].max())) + 1
    cfg = O.default_config(news_encoder='CNE', user_encoder='SUE', dataset='small', vocabulary_size=V, word_embedding_dim=16, hidden_dim=8,
                           attention_dim=8, max_history_num=int(c['max_history_num']), max_title_length=8, max_abstract_length=16,
                           category_num=int(c['category_num']), subCategory_num=int(c['news_subCategory'].max()) + 1, category_embedding_dim=4,
                           subCategory_embedding_dim=4, negative_sample_num=4, head_num=2, head_dim=4, cnn_kernel_num=12, gcn_layer_num=2,
                           dropout_rate=0.0, lr=1e-2, user_num=int(c['beh_user'].max()) + 1, tie_order='stable', batch_size=4)
    torch.manual_seed(3)
    ref = O.Model(cfg)
    ref.initialize()
    with torch.no_grad():
        for p in ref.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0, 0.05)
            p.mul_(1.5)
    ref.train()
    model = Model(cfg, torch.zeros(V, 16))
    model.load_state_dict(ref.state_dict())
    model = model.cuda().train()
    trainer = Trainer(model, cfg)
    opt = O.make_optimizer(ref, cfg)
    dc = DeviceCorpus(c, 'cuda', int(c['category_num']), graph='build', norm='symmetric')
    dc.set_samples(c['train_samples'])
    n = int(c['beh_user'].shape[0])
    order = dp.sampler_indices(n, 0, 1, epoch=0).numpy()                                   # DistributedSampler(seed 0), epoch 0, world 1
    worst = 0.0
    for start in range(0, n - 3, 4):
        idx = order[start:start + 4].astype(np.int32)
        _, loss = trainer.train_step(dc.train_batch(idx))
        ref_batch = [torch.from_numpy(np.ascontiguousarray(a)) for a in CO.train_batch(c, idx)]
        _, ref_loss = O.train_step(ref, opt, ref_batch, cfg.gradient_clip_norm)
        worst = max(worst, abs(float(loss) - float(ref_loss)))
    print('epoch of %d steps: worst |loss - oracle loss| = %.2e' % (n // 4, worst))
    assert worst <= 5e-5
    rp = dict(ref.named_parameters())
    for k, p in model.named_parameters():
        if k.startswith('user_encoder.news_encoder.'):
            continue
        assert float((p.detach().cpu() - rp[k].detach()).abs().max()) <= 4 * float(cfg.lr) + 1e-4, k     # Adam: <= lr per step per element
