#!/usr/bin/env python3
"""Solo timing of the embedding-row gradient forms on the headline shapes: f32-atomic scatter vs token sort + sorted segmented
reduction (csrc/sort.hip).  Zipf word ids as the synthetic corpus draws them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nnr_amd import ops
from nnr_amd.synth import SynthSpec, SynthCorpus

d = torch.device('cuda')
V, E = 60000, 300
spec = SynthSpec(vocabulary_size=V)
corpus = SynthCorpus(spec)
BS = int(sys.argv[sys.argv.index('--batch_size') + 1]) if '--batch_size' in sys.argv else 64
b = corpus.batch(BS, np.random.default_rng(100))


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1000 * e0.elapsed_time(e1) / n


for name, key_t, key_m in (('content', 'content_text', 'content_mask'), ('title', 'title_text', 'title_mask')):
    ids = np.concatenate([b['news_' + key_t].reshape(-1, b['news_' + key_t].shape[-1]), b['user_' + key_t].reshape(-1, b['user_' + key_t].shape[-1])])
    mask = np.concatenate([b['news_' + key_m].reshape(ids.shape[0] - b['user_' + key_t].reshape(-1, ids.shape[1]).shape[0], -1), b['user_' + key_m].reshape(-1, ids.shape[1])])
    mask[:, 0] = True
    live_ids = ids[mask.astype(bool)]
    live = live_ids.size
    cap = ids.size
    tok = torch.zeros(cap, dtype=torch.int32)
    tok[:live] = torch.from_numpy(live_ids.astype(np.int32))
    tok = tok.to(d)
    total = torch.tensor([live], dtype=torch.int32, device=d)
    dout = torch.randn(cap, E, device=d)
    table = torch.zeros(V, E, device=d)
    t_atomic = timed(lambda: ops.embed_scatter(dout, tok, table, 0.2, 5, dyn=total))
    t_sort = timed(lambda: ops.TokenSort(tok, total, V))
    ts = ops.TokenSort(tok, total, V)
    torch.cuda.synchronize()
    t_seg = timed(lambda: ops.embed_scatter_sorted(dout, ts, table, 0.2, 5))
    uniq = int(torch.unique(tok[:live]).numel())
    print('%-8s cap %7d live %6d unique words %6d: atomic scatter %7.1f us | token sort %7.1f us (leaf stream, forward) + segmented reduction %7.1f us (%.0f GB/s of gradient rows)'
          % (name, cap, live, uniq, t_atomic, t_sort, t_seg, live * E * 4 / t_seg / 1e3))
