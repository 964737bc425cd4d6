"""Synthetic MIND-shaped batches.

MIND itself cannot be present on the bench box, so the bench, the parity tests and the golden
generator draw from this deterministic generator.  It reproduces the *contract* of the reference's
DataLoader (`MIND_dataset.py:70-76`: 21 tensors, dtypes and shapes of Appendix C) and the corpus rules
that shape those tensors:
  * news 0 is the <PAD> news: all-zero ids, mask [1,0,...] (MIND_corpus.py:352-353), category 0;
  * token masks are prefix-shaped, ids are zero past the length (MIND_corpus.py:308-340);
  * history = last <=50 clicked news, right-padded with news 0 (MIND_corpus.py:367-371);
  * user_history_graph / category mask / category indices follow MIND_corpus.py:179-216
    (identity + intra-category cliques + news<->proxy + proxy<->proxy edges, D^-1/2 A D^-1/2,
    left un-normalised when the history is empty).
Vocabulary size and length moments are generator parameters (assumptions from the public MIND
description, not from the reference) and are reported with every bench line.
"""
from dataclasses import dataclass, asdict

import numpy as np

BATCH_FIELDS = (
    'user_ID', 'user_category', 'user_subCategory', 'user_title_text', 'user_title_mask', 'user_title_entity',
    'user_content_text', 'user_content_mask', 'user_content_entity', 'user_history_mask', 'user_history_graph',
    'user_history_category_mask', 'user_history_category_indices', 'news_category', 'news_subCategory',
    'news_title_text', 'news_title_mask', 'news_title_entity', 'news_content_text', 'news_content_mask',
    'news_content_entity')


@dataclass
class SynthSpec:
    vocabulary_size: int = 60000
    category_num: int = 18
    subCategory_num: int = 285
    max_title_length: int = 32
    max_abstract_length: int = 128
    max_history_num: int = 50
    negative_sample_num: int = 4
    news_pool: int = 20000
    title_len_mean: float = 11.5
    content_len_mean: float = 43.0
    len_sigma: float = 0.45
    empty_content_frac: float = 0.05
    empty_history_frac: float = 0.02
    zipf_s: float = 1.1
    dense: bool = False          # all lengths = max (worst-case roofline variant)
    seed: int = 1

    def describe(self):
        return asdict(self)


def _zipf_ids(rng, size, lo, hi, s):
    """Zipf(s)-distributed integers in [lo, hi) by inverse-CDF on a truncated power law."""
    n = hi - lo
    ranks = np.arange(1, n + 1, dtype=np.float64)
    cdf = np.cumsum(ranks ** (-s))
    cdf /= cdf[-1]
    u = rng.random(size)
    return (np.searchsorted(cdf, u).astype(np.int64) + lo).astype(np.int32)


def _lengths(rng, size, mean, sigma, lo, hi):
    mu = np.log(mean) - 0.5 * sigma * sigma
    return np.clip(np.rint(rng.lognormal(mu, sigma, size)), lo, hi).astype(np.int64)


class SynthCorpus:
    """A pool of synthetic news plus a batch sampler with the reference DataLoader's output contract."""

    def __init__(self, spec: SynthSpec):
        self.spec = spec
        rng = np.random.default_rng(spec.seed)
        P, T, C = spec.news_pool, spec.max_title_length, spec.max_abstract_length
        if spec.dense:
            tl = np.full(P, T, dtype=np.int64)
            cl = np.full(P, C, dtype=np.int64)
        else:
            tl = _lengths(rng, P, spec.title_len_mean, spec.len_sigma, 1, T)
            cl = _lengths(rng, P, spec.content_len_mean, spec.len_sigma, 1, C)
            cl[rng.random(P) < spec.empty_content_frac] = 0
        tl[0] = 0
        cl[0] = 0
        self.title_len, self.content_len = tl, cl
        pos_t = np.arange(T)[None, :]
        pos_c = np.arange(C)[None, :]
        self.title_mask = pos_t < tl[:, None]
        self.content_mask = pos_c < cl[:, None]
        self.title_text = np.where(self.title_mask, _zipf_ids(rng, (P, T), 2, spec.vocabulary_size, spec.zipf_s), 0).astype(np.int32)
        self.content_text = np.where(self.content_mask, _zipf_ids(rng, (P, C), 2, spec.vocabulary_size, spec.zipf_s), 0).astype(np.int32)
        # <PAD>-news convention: mask position 0 is set (MIND_corpus.py:352-353)
        self.title_mask[0, 0] = True
        self.content_mask[0, 0] = True
        self.title_entity = np.zeros((P, T), dtype=np.int32)      # unused by CNE / MHSA / CNN
        self.content_entity = np.zeros((P, C), dtype=np.int32)
        self.category = _zipf_ids(rng, P, 0, spec.category_num, 1.0)
        self.subCategory = _zipf_ids(rng, P, 0, spec.subCategory_num, 1.0)
        self.category[0] = 0
        self.subCategory[0] = 0
        self._rng = rng

    # ------------------------------------------------------------------ graph rule (MIND_corpus.py:179-216)
    def history_graph(self, cats, count):
        s = self.spec
        H, K = s.max_history_num, s.category_num
        G = H + K
        A = np.identity(G, dtype=np.float32)
        cmask = np.zeros(K + 1, dtype=bool)
        cidx = np.full(H, K, dtype=np.int64)
        if count > 0:
            c = cats[:count].astype(np.int64)
            cidx[:count] = c
            cmask[c] = True
            i = np.arange(count)
            A[i, H + c] = 1
            A[H + c, i] = 1
            same = c[:, None] == c[None, :]
            A[:count, :count][same] = 1
            diff_pairs = np.argwhere(~same)
            A[H + c[diff_pairs[:, 0]], H + c[diff_pairs[:, 1]]] = 1
            d = np.sqrt(1.0 / A.sum(axis=1)).astype(np.float32)
            A = (d[:, None] * A) * d[None, :]
        return A.astype(np.float32), cmask, cidx

    # ------------------------------------------------------------------ batches
    def batch(self, batch_size, rng=None):
        """One batch as a dict of numpy arrays, keys/order = BATCH_FIELDS (= Model.forward's positional order)."""
        s = self.spec
        rng = self._rng if rng is None else rng
        B, H, N, K = batch_size, s.max_history_num, 1 + s.negative_sample_num, s.category_num
        counts = rng.integers(0, H + 1, size=B)
        counts[rng.random(B) < s.empty_history_frac] = 0
        hist = np.zeros((B, H), dtype=np.int64)
        hmask = np.zeros((B, H), dtype=bool)
        for b in range(B):
            hist[b, :counts[b]] = rng.integers(1, s.news_pool, size=counts[b])
            hmask[b, :counts[b]] = True
        cand = rng.integers(1, s.news_pool, size=(B, N))
        graph = np.zeros((B, H + K, H + K), dtype=np.float32)
        cmask = np.zeros((B, K + 1), dtype=bool)
        cidx = np.zeros((B, H), dtype=np.int64)
        for b in range(B):
            graph[b], cmask[b], cidx[b] = self.history_graph(self.category[hist[b]], int(counts[b]))
        out = {
            'user_ID': np.arange(B, dtype=np.int64),
            'user_category': self.category[hist], 'user_subCategory': self.subCategory[hist],
            'user_title_text': self.title_text[hist], 'user_title_mask': self.title_mask[hist],
            'user_title_entity': self.title_entity[hist],
            'user_content_text': self.content_text[hist], 'user_content_mask': self.content_mask[hist],
            'user_content_entity': self.content_entity[hist],
            'user_history_mask': hmask, 'user_history_graph': graph,
            'user_history_category_mask': cmask, 'user_history_category_indices': cidx,
            'news_category': self.category[cand], 'news_subCategory': self.subCategory[cand],
            'news_title_text': self.title_text[cand], 'news_title_mask': self.title_mask[cand],
            'news_title_entity': self.title_entity[cand],
            'news_content_text': self.content_text[cand], 'news_content_mask': self.content_mask[cand],
            'news_content_entity': self.content_entity[cand],
        }
        return {k: np.ascontiguousarray(out[k]) for k in BATCH_FIELDS}


def to_torch(batch, device='cpu'):
    """dict of numpy arrays -> list of 21 contiguous torch tensors in Model.forward order."""
    import torch
    return [torch.from_numpy(np.array(batch[k], copy=True)).to(device) for k in BATCH_FIELDS]
