"""oracle/eval_oracle.py against the reference's own evaluate.py / util.compute_scores outputs (tests/golden/eval_*.npz)."""
import os

import numpy as np
import pytest

from oracle import eval_oracle as EO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_metrics_match_reference_evaluate_py():
    z = np.load(os.path.join(GOLD, 'eval_metrics_ragged.npz'))
    per, mean = EO.scoring(z['labels'], z['ranks'], z['sizes'])
    np.testing.assert_allclose(per, z['per_impression'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(mean, z['metrics'], rtol=0, atol=1e-12)


@pytest.mark.parametrize('tag', ['tiny_MHSA_MHSA', 'tiny_CNN_ATT', 'tiny_CNE_SUE_stable'])
def test_ranks_and_metrics_match_reference_compute_scores(tag):
    z = np.load(os.path.join(GOLD, 'eval_%s.npz' % tag))
    ranks = EO.ranks_from_scores(z['scores'], z['sizes'])
    np.testing.assert_array_equal(ranks, z['ranks'])
    _, mean = EO.scoring(z['labels'], ranks, z['sizes'])
    np.testing.assert_allclose(mean, z['metrics'], rtol=0, atol=1e-12)
    # samples of one impression are contiguous and in file order (util.py:50-52)
    assert np.array_equal(np.repeat(np.arange(len(z['sizes'])), z['sizes']), z['dev_indices'])


def test_ties_keep_file_order():
    r = EO.ranks_from_scores(np.array([0.5, 0.7, 0.5, 0.7, 0.1], dtype=np.float32), [5])
    assert r.tolist() == [3, 1, 4, 2, 5]
