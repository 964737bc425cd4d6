#!/usr/bin/env python3
"""One NT GEMM shape, a few tiles, a few launches each -- the workload of a `rocprofv3 --pmc ...` pass (summarised by
tools/pmc_summary.py).  Usage: gemm_pmc.py M N K tile[,tile...] [rocblas]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops
d = torch.device('cuda')
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tiles = [int(t) for t in sys.argv[4].split(',')]
a = torch.randn(M, K, device=d); b = torch.randn(N, K, device=d) * 0.05; c = torch.empty(M, N, device=d)
for _ in range(4):
    for t in tiles:
        ops.gemm(a, b, c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=t)
    if 'rocblas' in sys.argv:
        torch.mm(a, b.t(), out=c)
torch.cuda.synchronize()
