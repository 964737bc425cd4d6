#!/usr/bin/env python3
"""The additive-attention GEMM (tanh(x W1^T + b) with the fused w2 row-dot, layers.py:167-169) alone, at in-step shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops
from tools.gemm_bench import timeit
d = torch.device('cuda')
for cap, live in ((409600, 140000), (102400, 37000), (40960, 14000)):
    x = torch.randn(cap, 400, device=d); w1 = torch.randn(200, 400, device=d) * 0.05; b1 = torch.randn(200, device=d); w2 = torch.randn(1, 200, device=d)
    th = torch.empty(cap, 200, device=d); sc = torch.empty(cap, device=d)
    dyn = torch.tensor([live], device=d, dtype=torch.int32)
    ms = timeit(lambda: ops.gemm(x, w1, None, M=cap, N=200, K=400, lda=400, ldb=400, dyn=dyn, dyn_dim=1, bias=b1, act=ops.ACT_TANH, aux_out=th, ldaux=200,
                                 rowdot_w=w2, rowdot_out=sc, tile=3))
    print('rowdot fused  M %6d/%-6d: %7.1f us %5.1f TF' % (live, cap, ms * 1e3, 2.0 * live * 200 * 400 / ms / 1e9))
    for tile in (4, 5, 2):
        ms = timeit(lambda: ops.gemm(x, w1, th, M=cap, N=200, K=400, lda=400, ldb=400, ldc=200, dyn=dyn, dyn_dim=1, bias=b1, act=ops.ACT_TANH, tile=tile))
        print('   plain tanh GEMM tile %d:      %7.1f us %5.1f TF' % (tile, ms * 1e3, 2.0 * live * 200 * 400 / ms / 1e9))
