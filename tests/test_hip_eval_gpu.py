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
