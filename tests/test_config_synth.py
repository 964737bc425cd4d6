"""Host-side contract tests: flag namespace (reference config.py names / per-dataset overrides) and the synthetic
MIND-shaped batch generator (dtypes and shapes of the reference DataLoader; graph rule A.6)."""
import numpy as np
import torch

from nnr_amd.config import make_config
from nnr_amd.synth import SynthSpec, SynthCorpus, BATCH_FIELDS, to_torch


def test_flag_names_defaults_and_dataset_overrides():
    c = make_config([])
    assert (c.news_encoder, c.user_encoder, c.batch_size, c.lr, c.gradient_clip_norm) == ('CNE', 'SUE', 64, 1e-4, 4.0)
    assert (c.max_title_length, c.max_abstract_length, c.max_history_num, c.negative_sample_num) == (32, 128, 50, 4)
    assert (c.hidden_dim, c.attention_dim, c.head_num, c.head_dim, c.word_embedding_dim) == (200, 200, 20, 20, 300)
    assert (c.dropout_rate, c.gcn_layer_num, c.epoch) == (0.2, 4, 8)                     # 200k block, config.py:88-90
    s = make_config(['--dataset=small', '--dropout_rate=0.9'])
    assert (s.dropout_rate, s.gcn_layer_num) == (0.25, 3)                                # CLI value overwritten, config.py:84-87
    l = make_config(['--dataset=large', '--batch_size=128', '--world_size=8'])
    assert (l.dropout_rate, l.epoch, l.batch_size // l.world_size) == (0.1, 6, 16)


def test_batch_contract_matches_reference_dataloader():
    spec = SynthSpec(vocabulary_size=500, news_pool=200, seed=2)
    b = SynthCorpus(spec).batch(6, np.random.default_rng(0))
    assert tuple(b.keys()) == BATCH_FIELDS
    B, H, N, K, T, C = 6, 50, 5, 18, 32, 128
    want = {'user_ID': ((B,), np.int64), 'user_category': ((B, H), np.int32), 'user_title_text': ((B, H, T), np.int32),
            'user_title_mask': ((B, H, T), np.bool_), 'user_content_text': ((B, H, C), np.int32),
            'user_history_mask': ((B, H), np.bool_), 'user_history_graph': ((B, H + K, H + K), np.float32),
            'user_history_category_mask': ((B, K + 1), np.bool_), 'user_history_category_indices': ((B, H), np.int64),
            'news_category': ((B, N), np.int32), 'news_title_text': ((B, N, T), np.int32), 'news_content_mask': ((B, N, C), np.bool_)}
    for k, (shape, dt) in want.items():
        assert b[k].shape == shape and b[k].dtype == dt, k
    # masks are prefix-shaped, ids are zero past the length
    m, ids = b['user_content_mask'], b['user_content_text']
    assert np.all(m[..., 1:] <= m[..., :-1]) and np.all(ids[~m] == 0)
    for t in to_torch(b):
        assert t.is_contiguous()


def test_graph_rule():
    spec = SynthSpec(vocabulary_size=100, news_pool=50, category_num=4, max_history_num=6, seed=1)
    corp = SynthCorpus(spec)
    cats = np.array([2, 0, 2, 3, 0, 0])
    A, cmask, cidx = corp.history_graph(cats, 4)
    G = 6 + 4
    raw = np.identity(G)
    c = cats[:4]
    for i in range(4):
        raw[i, 6 + c[i]] = raw[6 + c[i], i] = 1
        for j in range(i + 1, 4):
            if c[i] == c[j]:
                raw[i, j] = raw[j, i] = 1
            else:
                raw[6 + c[i], 6 + c[j]] = raw[6 + c[j], 6 + c[i]] = 1
    d = np.sqrt(1 / raw.sum(1))
    np.testing.assert_allclose(A, d[:, None] * raw * d[None, :], rtol=1e-6)
    assert cidx.tolist() == [2, 0, 2, 3, 4, 4] and cmask.tolist() == [True, False, True, True, False]
    A0, cm0, ci0 = corp.history_graph(cats, 0)                       # empty history: identity, un-normalised
    np.testing.assert_array_equal(A0, np.identity(G, dtype=np.float32))
    assert not cm0.any() and (ci0 == 4).all()


def test_device_corpus_refuses_cpu_and_negative_sampling_rules():
    """The data side has no CPU path either; the host-side sampler follows MIND_dataset.py:27-47."""
    import numpy as np
    import pytest
    from nnr_amd import _lib
    from nnr_amd.corpus import DeviceCorpus, negative_sampling
    with pytest.raises(_lib.NnrHipError):
        DeviceCorpus({}, 'cpu', 18)
    rs = np.random.RandomState(0)
    s = negative_sampling([(9, [4]), (8, [1, 2, 3, 5, 6, 7])], 4, rs.randint)
    assert s.dtype == np.int32 and s[0].tolist() == [9, 4, 4, 4, 4]
    assert s[1, 0] == 8 and len(set(s[1, 1:].tolist())) == 4 and set(s[1, 1:].tolist()) <= {1, 2, 3, 5, 6, 7}


def test_gcn_graph_flags_follow_the_reference():
    """config.py:56-58,111: --gcn_normalization_type {symmetric, asymmetric}; --no_self_connection only with
    --no_adjacent_normalization; the device corpus derives its graph normalisation from these flags."""
    import pytest
    from nnr_amd.config import make_config
    from nnr_amd.corpus import norm_from_config
    assert make_config([]).gcn_normalization_type == 'symmetric'
    assert norm_from_config(make_config(['--gcn_normalization_type=asymmetric'])) == 'asymmetric'
    assert norm_from_config(make_config(['--no_adjacent_normalization'])) == 'none'
    assert norm_from_config(make_config(['--no_adjacent_normalization', '--no_self_connection'])) == 'none_noself'
    with pytest.raises(AssertionError):
        make_config(['--no_self_connection'])
    with pytest.raises(SystemExit):
        make_config(['--gcn_normalization_type=bogus'])
