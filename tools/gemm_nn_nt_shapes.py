#!/usr/bin/env python3
"""The row-parallel GEMMs (forward NT, backward-data NN) of one CNE+SUE step, each ALONE at its in-step shape (dynamic M)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops
from tools.gemm_bench import timeit
d = torch.device('cuda')
cases = [  # (label, kind, cap M, live M, N, K)
    ('x.W_ih  hist content', 'nt', 409600, 140000, 1664, 300), ('x.W_ih  hist title', 'nt', 102400, 37000, 1664, 300),
    ('x.W_ih  cand content', 'nt', 40960, 14000, 1664, 300), ('H-lin   hist content', 'nt', 409600, 140000, 400, 400),
    ('H-lin   hist title', 'nt', 102400, 37000, 400, 400), ('GCN XW  (SUE)', 'nt', 4352, 4352, 900, 900),
    ('dX emb  hist content', 'nn', 409600, 140000, 300, 1664), ('dX emb  hist title', 'nn', 102400, 37000, 300, 1664),
    ('dX emb  cand content', 'nn', 40960, 14000, 300, 1664), ('dH      hist content', 'nn', 409600, 140000, 400, 400),
    ('dHt attn hist content', 'nn', 409600, 140000, 400, 200), ('dH      hist title', 'nn', 102400, 37000, 400, 400),
    ('GCN dX  (SUE)', 'nn', 4352, 4352, 900, 900),
]
for label, kind, cap, live, N, K in cases:
    a = torch.randn(cap, K, device=d)
    b = torch.randn(N, K, device=d) if kind == 'nt' else torch.randn(K, N, device=d)
    c = torch.empty(cap, N, device=d)
    dyn = torch.tensor([live], device=d, dtype=torch.int32) if live < cap else None
    res = []
    for tile in (0, 1, 2, 4, 5):
        try:
            ms = timeit(lambda: ops.gemm(a, b, c, M=cap, N=N, K=K, lda=K, ldb=(K if kind == 'nt' else N), ldc=N, trans_b=(kind == 'nn'), dyn=dyn, dyn_dim=1, tile=tile))
            res.append('t%d %6.1f us %5.1f TF' % (tile, ms * 1e3, 2.0 * live * N * K / ms / 1e9))
        except Exception as e:
            res.append('t%d n/a' % tile)
    print('%-22s %s M %6d/%-6d N%-4d K%-4d | %s' % (label, kind, live, cap, N, K, ' | '.join(res)), flush=True)
    del a, b, c
