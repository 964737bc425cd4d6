"""Evaluation path on the device (nnr_amd.evaluate, csrc/corpus.hip:rank_metrics_kernel) against what the reference's own
util.compute_scores / evaluate.py produced (tests/golden/eval_*.npz).  Scores: fp32 within 1e-4 (north-star bar; 2e-5 expected);
ranks: exact; metrics: float64 within 1e-12."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_rank_metrics_kernel_matches_reference_evaluate_py():
    from nnr_amd.evaluate import rank_metrics
    z = np.load(os.path.join(GOLD, 'eval_metrics_ragged.npz'))
    scores = torch.from_numpy((1.0 / z['ranks']).astype(np.float32)).cuda()      # what evaluate.py itself scores with (evaluate.py:66-73)
    ranks, per, mean = rank_metrics(scores, torch.from_numpy(z['labels']), z['sizes'])
    np.testing.assert_array_equal(ranks.cpu().numpy(), z['ranks'])
    np.testing.assert_allclose(per.cpu().numpy(), z['per_impression'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(mean.cpu().numpy(), z['metrics'], rtol=0, atol=1e-12)


def test_rank_metrics_ties_and_single_class():
    from nnr_amd.evaluate import rank_metrics
    from oracle import eval_oracle as EO
    s = torch.tensor([0.5, 0.7, 0.5, 0.7, 0.1, 1.0, 1.0, 0.3, 0.2], dtype=torch.float32).cuda()
    y = torch.tensor([1, 0, 0, 1, 0, 1, 1, 0, 0], dtype=torch.uint8)
    ranks, per, _ = rank_metrics(s, y, [5, 2, 2])
    assert ranks.cpu().tolist() == [3, 1, 4, 2, 5, 1, 2, 1, 2]                   # equal scores keep file order (stable sort, util.py:55)
    per = per.cpu().numpy()
    np.testing.assert_allclose(per[0], EO.impression_metrics([1, 0, 0, 1, 0], [3, 1, 4, 2, 5]), atol=1e-12)
    assert np.isnan(per[1]).all() and np.isnan(per[2]).all()                      # all clicked / none clicked: sklearn's roc_auc_score raises


@pytest.mark.parametrize('tag', ['tiny_MHSA_MHSA', 'tiny_CNN_ATT', 'tiny_CNE_SUE_stable'])
@pytest.mark.parametrize('graph', ['build', 'table'])
def test_compute_scores_and_metrics_match_reference(tag, graph):
    from nnr_amd.evaluate import dev_corpus, compute_scores, rank_metrics
    from nnr_amd.model import Model
    z = np.load(os.path.join(GOLD, 'eval_%s.npz' % tag))
    cast = {'int': int, 'float': float, 'str': str, 'bool': lambda v: v == 'True'}
    cfg = SimpleNamespace(**{k: cast[t](v) for k, v, t in zip(z['cfg_keys'], z['cfg_vals'], z['cfg_types'])})
    cfg.tie_order = str(z['tie_order'])
    model = Model(cfg, torch.zeros(cfg.vocabulary_size, cfg.word_embedding_dim))
    state = {k[len('state/'):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('state/')}
    model.load_state_dict(state)                                                  # the reference's own state_dict, names unchanged
    model = model.cuda()
    model.train()                                                                 # compute_scores must switch to eval itself (dropout 0.2)
    dc = dev_corpus({k: z[k] for k in z.files}, 'cuda', int(z['category_num']), graph=graph)
    scores = compute_scores(model, dc, batch_size=8)
    assert model.training
    got = scores.cpu().numpy()
    err = float(np.abs(got - z['scores']).max())
    print('%s scores max-abs-err %.3e' % (tag, err))
    assert err <= 2e-5, err
    ranks, per, mean = rank_metrics(scores, torch.from_numpy(z['labels']), z['sizes'])
    gaps = []
    o = 0
    for n in z['sizes']:
        s = np.sort(z['scores'][o:o + n]); gaps.append(np.diff(s).min() if n > 1 else 1.0); o += n
    if min(gaps) > 1e-4:                                                          # ranking is then unambiguous at the parity tolerance
        np.testing.assert_array_equal(ranks.cpu().numpy(), z['ranks'])
        np.testing.assert_allclose(mean.cpu().numpy(), z['metrics'], rtol=0, atol=1e-12)
    # ranks of the reference's own scores -> the reference's metrics, independent of model parity
    r2, _, m2 = rank_metrics(torch.from_numpy(z['scores']).cuda(), torch.from_numpy(z['labels']), z['sizes'])
    np.testing.assert_array_equal(r2.cpu().numpy(), z['ranks'])
    np.testing.assert_allclose(m2.cpu().numpy(), z['metrics'], rtol=0, atol=1e-12)


@pytest.mark.parametrize('tag', ['tiny_MHSA_MHSA', 'tiny_CNN_ATT'])
def test_cached_news_representations_give_the_reference_scores(tag):
    """f-3 (results-preserving part): for the MHSA / CNN encoders every distinct news is encoded ONCE per evaluation and gathered
    by id.  Same scores as the reference's per-sample util.compute_scores (goldens) and as this package's own per-sample form;
    the encoder processes several times fewer news rows.  CNE must refuse (rank-pairing quirk)."""
    from nnr_amd import evaluate as E
    from nnr_amd.model import Model
    z = np.load(os.path.join(GOLD, 'eval_%s.npz' % tag))
    cast = {'int': int, 'float': float, 'str': str, 'bool': lambda v: v == 'True'}
    cfg = SimpleNamespace(**{k: cast[t](v) for k, v, t in zip(z['cfg_keys'], z['cfg_vals'], z['cfg_types'])})
    cfg.tie_order = str(z['tie_order'])
    model = Model(cfg, torch.zeros(cfg.vocabulary_size, cfg.word_embedding_dim))
    model.load_state_dict({k[len('state/'):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('state/')})
    model = model.cuda().train()
    dc = E.dev_corpus({k: z[k] for k in z.files}, 'cuda', int(z['category_num']))
    assert E.news_reps_cacheable(model)
    cached = E.compute_scores(model, dc, batch_size=8)                    # 'auto' -> cached
    st = dict(E.LAST_STATS)
    plain = E.compute_scores(model, dc, batch_size=8, cache=False)
    assert st['mode'] == 'cached' and E.LAST_STATS['mode'] == 'per-sample' and model.training
    assert float((cached - plain).abs().max()) <= 2e-6
    assert float(np.abs(cached.cpu().numpy() - z['scores']).max()) <= 2e-5
    assert st['encoder_rows'] * 2 <= st['per_sample_rows']               # tiny fixture: 60 news vs 6-7 rows per sample; MIND dev: > 10x
    print('%s: cached %d encoder rows vs %d per-sample rows' % (tag, st['encoder_rows'], st['per_sample_rows']))


def test_cne_pad_slot_dedup_gives_the_reference_scores():
    """f-3, exact part for CNE: PAD history slots gated only by PAD-like partners are not encoded (one representative is); the scores
    are the reference's util.compute_scores goldens and equal the full per-sample form."""
    from nnr_amd import evaluate as E
    from nnr_amd.model import Model
    z = np.load(os.path.join(GOLD, 'eval_tiny_CNE_SUE_stable.npz'))
    cast = {'int': int, 'float': float, 'str': str, 'bool': lambda v: v == 'True'}
    cfg = SimpleNamespace(**{k: cast[t](v) for k, v, t in zip(z['cfg_keys'], z['cfg_vals'], z['cfg_types'])})
    cfg.tie_order = str(z['tie_order'])
    model = Model(cfg, torch.zeros(cfg.vocabulary_size, cfg.word_embedding_dim))
    model.load_state_dict({k[len('state/'):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('state/')})
    model = model.cuda().train()
    dc = E.dev_corpus({k: z[k] for k in z.files}, 'cuda', int(z['category_num']))
    dedup = E.compute_scores(model, dc, batch_size=8)
    st = dict(E.LAST_STATS)
    model.news_encoder.pad_dedup = False
    plain = E.compute_scores(model, dc, batch_size=8)
    assert E.LAST_STATS['pad_slots_skipped'] == 0 and st['pad_slots_skipped'] > 0, (st, E.LAST_STATS)
    assert float((dedup - plain).abs().max()) <= 2e-6
    assert float(np.abs(dedup.cpu().numpy() - z['scores']).max()) <= 2e-5
    print('CNE eval: %d of %d encoder rows skipped (PAD slots with PAD partners)' % (st['pad_slots_skipped'], st['per_sample_rows']))


def test_cne_representations_are_not_cacheable():
    from nnr_amd import evaluate as E
    from nnr_amd.config import make_config
    from nnr_amd.model import Model
    m = Model(make_config(['--news_encoder=CNE', '--user_encoder=SUE'], corpus_sizes=dict(vocabulary_size=50)))
    assert not E.news_reps_cacheable(m)


def test_cached_evaluation_on_a_mind_shaped_dev_split():
    """MIND-shaped sizes: 20 000 news, 8 000 (impression, candidate) samples with up to 50-news histories (MIND-small's dev split
    has 2.7 M samples over 42 k news): >= 10x fewer encoder rows, same scores."""
    from nnr_amd import evaluate as E
    from nnr_amd.config import make_config
    from nnr_amd.corpus import from_synth
    from nnr_amd.model import Model
    from nnr_amd.synth import SynthSpec, SynthCorpus
    cfg = make_config(['--news_encoder=MHSA', '--user_encoder=MHSA'], corpus_sizes=dict(vocabulary_size=5000))
    torch.manual_seed(0)
    model = Model(cfg)
    model.initialize()
    model = model.cuda()
    synth = SynthCorpus(SynthSpec(vocabulary_size=5000, news_pool=20000, seed=2))
    dc = from_synth(synth, 8000, np.random.default_rng(1), 'cuda')
    dc.set_samples(dc.samples[:, :1].cpu().numpy())          # dev / test samples carry ONE candidate (util.py:43-48)
    a = E.compute_scores(model, dc, 256)
    st = dict(E.LAST_STATS)
    b = E.compute_scores(model, dc, 256, cache=False)
    assert st['encoder_rows'] * 10 <= st['per_sample_rows'], st
    assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))
