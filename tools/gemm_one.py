#!/usr/bin/env python3
"""One GEMM shape repeated (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops
d = torch.device('cuda')
M, E, NP2 = 131072, 300, 1664
mode = sys.argv[1] if len(sys.argv) > 1 else 'nt'
x = torch.randn(M, E, device=d); w = torch.randn(NP2, E, device=d) * 0.05; out = torch.empty(M, NP2, device=d)
dg = torch.randn(M, NP2, device=d); dw = torch.zeros(NP2, E, device=d)
for _ in range(5):
    if mode == 'nt':
        ops.gemm(x, w, out, M=M, N=NP2, K=E, lda=E, ldb=E, ldc=NP2)
    elif mode == 'nn':
        ops.gemm(dg, w, x, M=M, N=E, K=NP2, lda=NP2, ldb=E, ldc=E, trans_b=True)
    else:
        ops.gemm(dg, x, dw, M=NP2, N=E, K=M, lda=NP2, ldb=E, ldc=E, trans_a=True, trans_b=True, split_k=28, atomic=True)
torch.cuda.synchronize()
