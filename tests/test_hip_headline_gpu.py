"""Parity at the HEADLINE size (BASELINE.json: batch 64, MIND-200k-shaped synthetic batches, vocabulary 60 000, full model
dimensions): the HIP path driven by the product Trainer -- every HIP stream, the leaf-stream weight gradients, the CU-pair
recurrence on a full chip, the flat-buffer clip+Adam -- against the CPU oracle on the SAME batch.  This is the only size at
which all four streams, leaf deferral and ~880 pair-recurrence workgroups are active at once; a cross-stream race that needs
hundreds of tiles in flight would not show at the batch 2-8 fixtures.  Train mode, tie_order 'stable'; dropout 0 AND dropout
ON (the benchmarked configuration) with the HIP generator's keep-masks injected into the oracle at every site (tests/hip_masks.py).  Bars: logits 1e-4 (BASELINE.json north_star), loss 2e-5, every parameter gradient within 1e-4 of the
gradient scale, gradient norms 1e-4 relative, parameters after the Adam step as in tests/test_oracle_golden.py.
Reference sites: trainer.py:105-120, model.py:120-133, newsEncoders.py:102-141, userEncoders.py:68-98 / 164-173."""
import numpy as np
import pytest
import torch

from nnr_amd.config import make_config
from nnr_amd.synth import SynthSpec, SynthCorpus, BATCH_FIELDS, to_torch

pytestmark = pytest.mark.gpu


def _pair(cfg, seed, train=True):
    from nnr_amd.model import Model
    from oracle import nnr_oracle as O
    O.BiLSTM.backend = 'aten'                          # ATen's packed LSTM: pinned to the goldens like the time loop, 3x faster
    torch.manual_seed(seed)
    table = torch.randn(cfg.vocabulary_size, cfg.word_embedding_dim) * 0.3
    table[0] = 0
    ref = O.Model(cfg, table)
    ref.initialize()
    with torch.no_grad():
        for p in ref.parameters():                     # zero-initialised tensors (proxy nodes, biases) carry signal too
            if float(p.abs().max()) == 0.0:
                p.normal_(0, 0.05)
    model = Model(cfg)
    model.load_state_dict(ref.state_dict())
    model = model.cuda()
    (ref.train(), model.train()) if train else (ref.eval(), model.eval())
    return model, ref


def _check_step(model, ref, cfg, batch, lr_steps=1, inject_masks=False):
    from nnr_amd import ops
    from nnr_amd.trainer import Trainer
    from oracle import nnr_oracle as O
    trainer = Trainer(model, cfg)
    ops.lstm_sync_timeouts(reset=True)
    dev_batch = to_torch(batch, 'cuda')
    if inject_masks:
        # dropout ON: the keep-mask of every dropout site of THIS call, from the HIP generator, goes into the oracle
        import hip_masks
        rates = hip_masks.inject(model, ref, dict(zip(BATCH_FIELDS, dev_batch)))
        p = float(cfg.dropout_rate)
        for k, r in rates.items():
            if k.startswith(('cat', 'sub', 'sue')):
                assert abs(r - (1 - p)) < 0.01, (k, r)
            elif k.startswith('gcn'):
                assert abs(r - (1 - p / 2)) < 0.01, (k, r)
    logits, loss = trainer.train_step(dev_batch)
    torch.cuda.synchronize()
    assert ops.lstm_sync_timeouts() == 0
    got_grads = {k: p.grad.detach().cpu().double() for k, p in model.named_parameters()}
    got_norm = trainer.grad_total_norm()
    # oracle: the same step
    opt = O.make_optimizer(ref, cfg)
    rl = ref(*to_torch(batch))
    rloss = O.negative_log_softmax(rl)
    opt.zero_grad()
    rloss.backward()
    ref_grads = {k: p.grad.detach().double().clone() for k, p in ref.named_parameters()}
    rnorm = float(torch.nn.utils.clip_grad_norm_(ref.parameters(), cfg.gradient_clip_norm))
    opt.step()
    err = float((logits.cpu() - rl.detach()).abs().max())
    assert err <= 1e-4, 'logits differ by %.3e' % err
    assert abs(float(loss) - float(rloss)) <= 2e-5, (float(loss), float(rloss))
    assert abs(got_norm - rnorm) <= 1e-4 * max(1.0, rnorm), (got_norm, rnorm)
    worst = 0.0
    for k, rg in ref_grads.items():
        g = got_grads[k]
        scale = max(1e-3, 0.05 * rnorm, float(rg.norm()))
        d = float((g - rg).abs().max())
        worst = max(worst, d / scale)
        assert d <= 1e-4 * scale, 'grad %s: %.3e vs scale %.3e' % (k, d, scale)
        assert abs(float(g.norm()) - float(rg.norm())) <= 1e-4 * scale, 'grad norm ' + k
    # one Adam step: elements with a well-resolved gradient move identically; the rest move by -+lr each (the first Adam step
    # is lr * sign(g) and the sign of a gradient at the fp32 noise floor is noise): at most 2 * lr apart
    lr = float(cfg.lr)
    rp = dict(ref.named_parameters())
    for k, p in model.named_parameters():
        a, e = p.detach().cpu().numpy(), rp[k].detach().numpy()
        gabs = ref_grads[k].abs().numpy()
        resolved = gabs > 0.05 * max(float(gabs.max()), 1e-30)
        d = np.abs(a - e)
        assert d[resolved].max(initial=0.0) <= 5e-5, 'param ' + k
        assert d.max(initial=0.0) <= 2 * lr * 1.01 + 5e-5, 'param ' + k
    return err, worst


def test_cne_sue_batch64_vocab60000_DROPOUT_ON_matches_oracle():
    """The configuration bench.py measures (BASELINE.json configs[2]: CNE+SUE, MIND-200k, batch 64, V = 60 000, gcn 4, dropout
    0.2 ON, train mode) pinned to the oracle end to end: all six dropout sites of the reference (newsEncoders.py:53,117-118,
    userEncoders.py:80,91, layers.py:319-322 -- p/2 between GCN layers, none after the last, one proxy mask per sample, the
    in-place dropout after relu(Wx+b)+x) run with the masks of the HIP generator; logits, loss, every parameter gradient, the
    clipped norm and the Adam step are compared.  A wrong seed / offset / index / scale at any site -- forward gather,
    weight-gradient loader or scatter epilogue -- fails here."""
    cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=64'],
                      corpus_sizes=dict(vocabulary_size=60000), tie_order='stable')
    assert abs(cfg.dropout_rate - 0.2) < 1e-12 and cfg.gcn_layer_num == 4          # config.py:87-90
    model, ref = _pair(cfg, seed=2)
    batch = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size)).batch(64, np.random.default_rng(102))
    err, worst = _check_step(model, ref, cfg, batch, inject_masks=True)
    print('CNE+SUE batch 64, dropout 0.2 ON: logits max|diff| %.2e, worst gradient deviation %.2e of its scale' % (err, worst))


def test_cne_sue_batch64_REPLAYED_step_dropout_on_matches_oracle():
    """The step bench.py times is a native REPLAY of the recorded launch sequence (nnr_amd/tape.py).  Three steps bring the trainer
    to that state (call by call, call by call, record); the oracle is then synchronised to the product's parameters and the FOURTH
    step -- replayed from the tape on a new batch, with that step's seeds -- is compared: logits, loss, every gradient, total norm."""
    import hip_masks
    from nnr_amd import ops
    from nnr_amd.trainer import Trainer
    from oracle import nnr_oracle as O
    cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=64'],
                      corpus_sizes=dict(vocabulary_size=60000), tie_order='stable')
    model, ref = _pair(cfg, seed=5)
    trainer = Trainer(model, cfg)
    corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size))
    rng = np.random.default_rng(105)
    ops.lstm_sync_timeouts(reset=True)
    for want in ('native', 'native', 'record'):
        trainer.train_step(to_torch(corpus.batch(64, rng), 'cuda'))
        assert trainer.last_path == want
    torch.cuda.synchronize()
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    batch = corpus.batch(64, rng)
    dev_batch = to_torch(batch, 'cuda')
    hip_masks.inject(model, ref, dict(zip(BATCH_FIELDS, dev_batch)))
    logits, loss = trainer.train_step(dev_batch)
    torch.cuda.synchronize()
    assert trainer.last_path == 'replay' and ops.lstm_sync_timeouts() == 0
    got_norm = trainer.grad_total_norm()
    rl = ref(*to_torch(batch))
    rloss = O.negative_log_softmax(rl)
    ref.zero_grad()
    rloss.backward()
    rnorm = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in ref.parameters())))
    err = float((logits.cpu() - rl.detach()).abs().max())
    assert err <= 1e-4 and abs(float(loss) - float(rloss)) <= 2e-5, (err, float(loss), float(rloss))
    assert abs(got_norm - rnorm) <= 1e-4 * max(1.0, rnorm), (got_norm, rnorm)
    rp = dict(ref.named_parameters())
    worst = 0.0
    for k, p in model.named_parameters():
        g, rg = p.grad.detach().cpu().double(), rp[k].grad.double()
        scale = max(1e-3, 0.05 * rnorm, float(rg.norm()))
        d = float((g - rg).abs().max())
        worst = max(worst, d / scale)
        assert d <= 1e-4 * scale, 'grad %s: %.3e vs scale %.3e' % (k, d, scale)
    print('CNE+SUE batch 64, dropout ON, REPLAYED step: logits max|diff| %.2e, worst gradient deviation %.2e, tape %s' %
          (err, worst, trainer.tapes[next(iter(trainer.tapes))].info()))


def test_cne_sue_batch64_inference_with_pad_dedup_matches_oracle():
    """f-3 (exact part for the headline encoder) at the headline size: eval-mode logits of a MIND-shaped batch (half of the 3 200
    history slots are PAD news) with the redundant PAD slots not encoded, against the oracle's full forward and against the full
    HIP forward."""
    cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=64'],
                      corpus_sizes=dict(vocabulary_size=60000), tie_order='stable')
    model, ref = _pair(cfg, seed=6, train=False)
    batch = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size)).batch(64, np.random.default_rng(106))
    with torch.no_grad():
        got = model(*to_torch(batch, 'cuda'))
        enc, of = model.news_encoder._dedup_stats
        model.news_encoder.pad_dedup = False
        full = model(*to_torch(batch, 'cuda'))
        want = ref(*to_torch(batch))
    torch.cuda.synchronize()
    assert of == 3200 and enc <= 0.62 * of, (enc, of)
    e1, e2 = float((got.cpu() - want).abs().max()), float((got - full).abs().max())
    assert e1 <= 1e-4 and e2 <= 1e-5, (e1, e2)
    print('inference, batch 64: %d of %d history sequences encoded; logits max|diff| vs oracle %.2e, vs the full HIP forward %.2e' % (enc, of, e1, e2))


def test_cne_sue_large_batch16_vocab130000_DROPOUT_ON_matches_oracle():
    """BASELINE.json configs[4]'s per-GPU shard (MIND-large: dropout 0.1, config.py:91-94; batch 128 over 8 GPUs = 16 per GPU,
    trainer.py:218; V = 130 000), dropout ON with injected masks."""
    cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=large', '--batch_size=128', '--world_size=8'],
                      corpus_sizes=dict(vocabulary_size=130000), tie_order='stable')
    assert abs(cfg.dropout_rate - 0.1) < 1e-12
    model, ref = _pair(cfg, seed=3)
    batch = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size)).batch(16, np.random.default_rng(103))
    err, worst = _check_step(model, ref, cfg, batch, inject_masks=True)
    print('CNE+SUE batch 16 (large), dropout 0.1 ON: logits max|diff| %.2e, worst gradient deviation %.2e' % (err, worst))


def test_mhsa_mhsa_batch64_DROPOUT_ON_matches_oracle():
    """BASELINE.json configs[1] (MHSA+MHSA, batch 64) with every dropout site on: word rows and attention output at 0.2
    (newsEncoders.py:193,196), category rows (:53), and the user encoder's hard-wired p = 0.5 (userEncoders.py:171)."""
    cfg = make_config(['--news_encoder=MHSA', '--user_encoder=MHSA', '--dataset=200k', '--batch_size=64'],
                      corpus_sizes=dict(vocabulary_size=60000))
    model, ref = _pair(cfg, seed=4)
    batch = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size)).batch(64, np.random.default_rng(104))
    err, worst = _check_step(model, ref, cfg, batch, inject_masks=True)
    print('MHSA+MHSA batch 64, dropout ON: logits max|diff| %.2e, worst gradient deviation %.2e' % (err, worst))
