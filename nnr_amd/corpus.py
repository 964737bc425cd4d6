"""Device-resident corpus + id-only batches: the MI355X counterpart of MIND_Train_Dataset
(MIND_dataset.py:9-79) and of the pre-computed user-history graphs (MIND_corpus.py:162-221).

The reference keeps the corpus tables in host numpy arrays, fancy-indexes 21 arrays per sample in Python
(MIND_dataset.py:70-76) and ships 98.6 KB per impression over PCIe every step (trainer.py:83-103).  Here the tables are
uploaded once; a batch is (behaviour indices, sampled news ids) and two kernels of libnnr_hip.so fill the 21 tensors in HBM:
`nnr_corpus_batch` (row gather) and `nnr_history_graph` (graph / cluster mask / cluster indices from the history's
category ids -- the [behaviour lines, G, G] fp32 table, 18.5 KB per line, never exists).  No CPU fallback."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .ops import _p, _s

NORM = {'none': 0, 'symmetric': 1, 'asymmetric': 2, 'none_noself': 3}


def norm_from_config(config):
    """The graph normalisation the reference's flags select (config.py:56-58,111; MIND_corpus.py:180-214):
    --no_self_connection (only legal together with --no_adjacent_normalization) -> zero diagonal, no normalisation;
    --no_adjacent_normalization -> none; else --gcn_normalization_type ('symmetric' | 'asymmetric')."""
    if getattr(config, 'no_self_connection', False):
        assert getattr(config, 'no_adjacent_normalization', False), 'Adjacent normalization of graph only can be set in case of self-connection'
        return 'none_noself'
    if getattr(config, 'no_adjacent_normalization', False):
        return 'none'
    return getattr(config, 'gcn_normalization_type', 'symmetric')

_NEWS_KEYS = ('news_category', 'news_subCategory', 'news_title_text', 'news_title_mask', 'news_title_entity',
              'news_abstract_text', 'news_abstract_mask', 'news_abstract_entity')


def negative_sampling(behaviors, negative_sample_num, randint):
    """MIND_Train_Dataset.negative_sampling (MIND_dataset.py:27-47), host side: `behaviors` = [(click, [non-clicks]), ...],
    `randint(lo, hi)` the generator the reference draws from (numpy.random.randint).  Returns int32 [n, 1 + K]."""
    out = np.zeros((len(behaviors), 1 + negative_sample_num), dtype=np.int32)
    for i, (click, negatives) in enumerate(behaviors):
        out[i, 0] = click
        news_num = len(negatives)
        if news_num <= negative_sample_num:
            for j in range(negative_sample_num):
                out[i, j + 1] = negatives[j % news_num]
        else:
            used = set()
            for j in range(negative_sample_num):
                while True:
                    k = randint(0, news_num)
                    if k not in used:
                        out[i, j + 1] = negatives[k]
                        used.add(k)
                        break
    return out


class DeviceCorpus:
    """Corpus tables in HBM.  `arrays`: news_* tables [news(, T|C)], beh_user [n] int64, beh_history [n, H] int32 (news
    indices, 0 = PAD news), beh_history_mask [n, H] bool, optionally beh_line [n] + train_user_history_graph /
    _category_mask / _category_indices (the reference's pre-built tables; used when graph='table')."""

    def __init__(self, arrays, device, category_num, graph='build', norm='symmetric', config=None):
        if config is not None:                       # the reference's flags decide (they cannot disagree with a separate argument)
            norm = norm_from_config(config)
        assert graph in ('build', 'table') and norm in NORM
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise L.NnrHipError('DeviceCorpus needs a GPU: the batch kernels live in libnnr_hip.so (no CPU path)')
        self.device, self.graph_mode, self.norm, self.category_num = dev, graph, NORM[norm], int(category_num)
        self.norm_name = norm
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
        t = {}
        for k in _NEWS_KEYS:
            t[k] = up(arrays[k], torch.bool if k.endswith('mask') else torch.int32)
        t['beh_user'] = up(arrays['beh_user'], torch.int64)
        t['beh_history'] = up(arrays['beh_history'], torch.int32)
        t['beh_history_mask'] = up(arrays['beh_history_mask'], torch.bool)
        self.T, self.Cn = t['news_title_text'].shape[1], t['news_abstract_text'].shape[1]
        self.H = t['beh_history'].shape[1]
        self.G, self.K1 = self.H + self.category_num, self.category_num + 1
        if graph == 'table':
            t['beh_line'] = up(arrays['beh_line'], torch.int32)
            t['graph_table'] = up(arrays['train_user_history_graph'], torch.float32)
            t['cmask_table'] = up(arrays['train_user_history_category_mask'], torch.bool)
            t['cidx_table'] = up(arrays['train_user_history_category_indices'], torch.int64)
            assert t['graph_table'].shape[1] == self.G
        self.t = t
        self.num = t['beh_user'].shape[0]
        self.samples = None
        ct = L.CorpusTables()
        ct.news_category, ct.news_subCategory = _p(t['news_category']), _p(t['news_subCategory'])
        ct.title_text, ct.title_mask, ct.title_entity = _p(t['news_title_text']), _p(t['news_title_mask']), _p(t['news_title_entity'])
        ct.abstract_text, ct.abstract_mask, ct.abstract_entity = _p(t['news_abstract_text']), _p(t['news_abstract_mask']), _p(t['news_abstract_entity'])
        ct.beh_user, ct.beh_history, ct.beh_history_mask = _p(t['beh_user']), _p(t['beh_history']), _p(t['beh_history_mask'])
        ct.beh_line, ct.graph_table, ct.cmask_table, ct.cidx_table = _p(t.get('beh_line')), _p(t.get('graph_table')), _p(t.get('cmask_table')), _p(t.get('cidx_table'))
        ct.T, ct.C, ct.H, ct.G, ct.K1 = self.T, self.Cn, self.H, self.G, self.K1
        self._ct = ct

    def set_samples(self, samples):
        """[behaviours, 1 + K] news ids: column 0 the clicked news, the rest the sampled non-clicks (`negative_sampling`)."""
        self.samples = torch.as_tensor(np.ascontiguousarray(samples), dtype=torch.int32).to(self.device)
        assert self.samples.shape[0] == self.num

    def resident_bytes(self):
        return sum(v.numel() * v.element_size() for v in self.t.values()) + (self.samples.numel() * 4 if self.samples is not None else 0)

    def train_batch(self, beh_idx):
        """The 21 tensors of trainer.py:105-106 for the behaviours `beh_idx` (int32 tensor on the device, or a sequence)."""
        if self.samples is None:
            raise L.NnrHipError('call set_samples() (negative sampling) before train_batch()')
        idx = beh_idx if torch.is_tensor(beh_idx) else torch.as_tensor(np.asarray(beh_idx), dtype=torch.int32)
        idx = idx.to(device=self.device, dtype=torch.int32).contiguous()
        B, S, H, T, Cn, G, K1, dev = idx.numel(), self.samples.shape[1], self.H, self.T, self.Cn, self.G, self.K1, self.device
        e = lambda shape, dt: torch.empty(shape, device=dev, dtype=dt)
        out = [e((B,), torch.int64),
               e((B, H), torch.int32), e((B, H), torch.int32), e((B, H, T), torch.int32), e((B, H, T), torch.bool), e((B, H, T), torch.int32),
               e((B, H, Cn), torch.int32), e((B, H, Cn), torch.bool), e((B, H, Cn), torch.int32),
               e((B, H), torch.bool), e((B, G, G), torch.float32), e((B, K1), torch.bool), e((B, H), torch.int64),
               e((B, S), torch.int32), e((B, S), torch.int32), e((B, S, T), torch.int32), e((B, S, T), torch.bool), e((B, S, T), torch.int32),
               e((B, S, Cn), torch.int32), e((B, S, Cn), torch.bool), e((B, S, Cn), torch.int32)]
        bo = L.BatchOut()
        for (name, _), ten in zip(L.BatchOut._fields_, out):
            setattr(bo, name, _p(ten))
        L.check(L.lib().nnr_corpus_batch(C.byref(self._ct), C.byref(bo), _p(idx), _p(self.samples), S, B, S, _s()), 'nnr_corpus_batch')
        if self.graph_mode == 'build':
            L.check(L.lib().nnr_history_graph(_p(out[1]), _p(out[9]), B, H, self.category_num, self.norm, _p(out[10]), _p(out[11]), _p(out[12]),
                                              _s()), 'nnr_history_graph')
        return out


def history_graph(cats, hmask, category_num, norm='symmetric'):
    """Graph / cluster mask / cluster indices for a batch from category ids [B, H] int32 and the history mask [B, H] bool."""
    B, H = cats.shape
    G = H + category_num
    graph = torch.empty((B, G, G), device=cats.device, dtype=torch.float32)
    cmask = torch.empty((B, category_num + 1), device=cats.device, dtype=torch.bool)
    cidx = torch.empty((B, H), device=cats.device, dtype=torch.int64)
    L.check(L.lib().nnr_history_graph(_p(cats.contiguous()), _p(hmask.contiguous()), B, H, category_num, NORM[norm], _p(graph), _p(cmask), _p(cidx),
                                      _s()), 'nnr_history_graph')
    return graph, cmask, cidx


def from_synth(synth, n_behaviors, rng, device, graph='build'):
    """A DeviceCorpus over a SynthCorpus news pool with `n_behaviors` synthetic behaviours (bench / tests)."""
    s = synth.spec
    H, S = s.max_history_num, 1 + s.negative_sample_num
    counts = rng.integers(0, H + 1, size=n_behaviors)
    counts[rng.random(n_behaviors) < s.empty_history_frac] = 0
    hist = np.zeros((n_behaviors, H), dtype=np.int32)
    hmask = np.arange(H)[None, :] < counts[:, None]
    for b in range(n_behaviors):
        hist[b, :counts[b]] = rng.integers(1, s.news_pool, size=counts[b])
    arrays = dict(news_category=synth.category, news_subCategory=synth.subCategory, news_title_text=synth.title_text,
                  news_title_mask=synth.title_mask, news_title_entity=synth.title_entity, news_abstract_text=synth.content_text,
                  news_abstract_mask=synth.content_mask, news_abstract_entity=synth.content_entity,
                  beh_user=np.arange(n_behaviors, dtype=np.int64), beh_history=hist, beh_history_mask=hmask)
    if graph == 'table':
        g = [synth.history_graph(synth.category[hist[b]], int(counts[b])) for b in range(n_behaviors)]
        arrays.update(beh_line=np.arange(n_behaviors, dtype=np.int32), train_user_history_graph=np.stack([x[0] for x in g]),
                      train_user_history_category_mask=np.stack([x[1] for x in g]), train_user_history_category_indices=np.stack([x[2] for x in g]))
    dc = DeviceCorpus(arrays, device, s.category_num, graph=graph)
    dc.set_samples(rng.integers(1, s.news_pool, size=(n_behaviors, S)).astype(np.int32))
    return dc
