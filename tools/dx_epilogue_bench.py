#!/usr/bin/env python3
"""Embedding-row gradient GEMM (77 k live rows x 300 x 1664, tile 9) under epilogue variants: what the scatter / atomics / dropout cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops
d = torch.device('cuda')
live, cap, N, K = 77000, 450560, 300, 1664
a = torch.randn(cap, K, device=d); b = torch.randn(N, K, device=d) * 0.05
dyn = torch.tensor([live], device=d, dtype=torch.int32)
table = torch.zeros(60000, N, device=d)
dense = torch.empty(cap, N, device=d)
idx = (torch.rand(cap, device=d) ** 3 * 59999).int()
ident = torch.arange(cap, device=d, dtype=torch.int32)


def t(fn, iters=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters)
    return best


base = dict(M=cap, N=N, K=K, lda=K, ldb=K, ldc=N, dyn=dyn, dyn_dim=1)
for tile in (9, 15):
    variants = {
        'plain store': lambda: ops.gemm(a, b, dense, tile=tile, **base),
        'atomic, identity rows': lambda: ops.gemm(a, b, dense, tile=tile, c_idx=ident, atomic=True, **base),
        'atomic scatter (zipf rows)': lambda: ops.gemm(a, b, table, tile=tile, c_idx=idx, atomic=True, **base),
        'atomic scatter + dropout': lambda: ops.gemm(a, b, table, tile=tile, c_idx=idx, atomic=True, drop=(4, 0.2, 7, N), **base),
    }
    for k, f in variants.items():
        ms = t(f)
        print('tile %2d  %-28s %7.1f us  %6.1f TF' % (tile, k, ms * 1e3, 2.0 * live * N * K / ms / 1e9), flush=True)
