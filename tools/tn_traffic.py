#!/usr/bin/env python3
"""Workload of a rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass: the dW_ih weight-gradient GEMM (1664 x 300 over 70 000 live token rows
of a 409 600-row buffer) on the LDS-DMA TN tile, a few launches.  Operand bytes per launch: 70 000 x (1664 + 300) x 4 = 550 MB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops
d = torch.device('cuda')
M, N, live, cap = 1664, 300, 70000, 409600
a = torch.randn(cap, M, device=d); b = torch.randn(cap, N, device=d) * 0.05; c = torch.zeros(M, N, device=d)
dyn = torch.tensor([live], device=d, dtype=torch.int32)
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 26
sk = ops.split_for(M, N, cap, 128, 80, 2048)
for _ in range(6):
    ops.gemm(a, b, c, M=M, N=N, K=cap, lda=M, ldb=N, ldc=N, trans_a=True, trans_b=True, split_k=sk, atomic=True, tile=tile, dyn=dyn, dyn_dim=2)
torch.cuda.synchronize()
