#!/usr/bin/env python3
"""The token-reduction (dW) GEMMs of one CNE+SUE step, each ALONE on the GPU at its in-step shape (dynamic K = live tokens)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops
from tools.gemm_bench import timeit
d = torch.device('cuda')
cases = [  # (label, M, N, cap K, live K)
    ('dW_ih  hist content', 1664, 300, 409600, 140000), ('dW_ih  hist title', 1664, 300, 102400, 37000),
    ('dW_ih  cand content', 1664, 300, 40960, 14000), ('dW_ih  cand title', 1664, 300, 10240, 3700),
    ('dW_hh  hist content', 832, 200, 409600, 140000), ('dW_hh  hist title', 832, 200, 102400, 37000),
    ('dW_hh  cand content', 832, 200, 40960, 14000), ('dW_hh  cand title', 832, 200, 10240, 3700),
    ('attn/gate hist content', 400, 400, 409600, 140000), ('attn/gate hist title', 400, 400, 102400, 37000),
    ('attn1 hist content', 200, 400, 409600, 140000), ('GCN dW', 900, 900, 4352, 4352),
]
tot = 0.0
for label, M, N, cap, live in cases:
    a = torch.randn(cap, M, device=d); b = torch.randn(cap, N, device=d); c = torch.zeros(M, N, device=d)
    dyn = torch.tensor([live], device=d, dtype=torch.int32)
    sk = ops.split_for(M, N, cap)
    res = []
    for tile in (0, 2, 4):
        try:
            ms = timeit(lambda: ops.gemm(a, b, c, M=M, N=N, K=cap, lda=M, ldb=N, ldc=N, trans_a=True, trans_b=True, split_k=sk, atomic=True,
                                         dyn=dyn if live < cap else None, dyn_dim=2, tile=tile))
            res.append('tile %d: %6.1f us %5.1f TF' % (tile, ms * 1e3, 2.0 * M * N * live / ms / 1e9))
        except Exception as e:
            res.append('tile %d: n/a' % tile)
    print('%-24s M%-5d N%-4d K %6d/%-6d split %3d | %s' % (label, M, N, live, cap, sk, ' | '.join(res)), flush=True)
    del a, b, c
