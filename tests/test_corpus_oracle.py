"""The data-side oracle (oracle/corpus_oracle.py) against what the reference's own MIND_corpus.py / MIND_dataset.py produced
(tests/golden/corpus_*.npz, tools/make_corpus_goldens.py).  Integer / byte work and fp32 normalisation: bit-exact."""
import os

import numpy as np
import pytest

from oracle import corpus_oracle as CO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TAGS = ['tiny_h50_sym', 'tiny_h8_asym', 'tiny_h8_none', 'tiny_h8_noself']


def load(tag):
    return dict(np.load(os.path.join(GOLD, 'corpus_%s.npz' % tag)))


@pytest.mark.parametrize('tag', TAGS)
def test_history_graph_matches_reference(tag):
    c = load(tag)
    H, K, norm = int(c['max_history_num']), int(c['category_num']), str(c['norm'])
    lens = c['line_history_len']
    assert lens.min() == 0 and lens.max() > 50            # empty, ragged and over-long histories are all in the fixture
    for line in range(lens.shape[0]):
        g, m, ix = CO.history_graph(c['line_history_category'][line], int(lens[line]), H, K, norm)
        np.testing.assert_array_equal(g, c['train_user_history_graph'][line])
        np.testing.assert_array_equal(m, c['train_user_history_category_mask'][line])
        np.testing.assert_array_equal(ix, c['train_user_history_category_indices'][line])
        assert g.dtype == np.float32 and ix.dtype == np.int64 and m.dtype == bool


@pytest.mark.parametrize('tag', TAGS)
def test_train_batch_matches_reference_dataloader(tag):
    c = load(tag)
    got = CO.train_batch(c, c['batch_index'])
    assert len(got) == 21
    for k, a in enumerate(got):
        exp = c['batch_%02d' % k]
        np.testing.assert_array_equal(a, exp, err_msg='batch field %d' % k)
        assert a.dtype == exp.dtype, (k, a.dtype, exp.dtype)
