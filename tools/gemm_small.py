#!/usr/bin/env python3
"""Small (latency-bound) GEMMs of the step, alone on the GPU: 64x80 BK=16 (tile 2) vs 64x80 BK=64 (tile 6)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops
from tools.gemm_bench import timeit
d = torch.device('cuda')
for label, kind, M, N, K in (('cand gate', 'nt', 320, 400, 400), ('cand Q', 'nt', 320, 200, 400), ('hist Q', 'nt', 3200, 200, 400), ('hist M-lin', 'nt', 3200, 400, 400),
                             ('SUE GCN', 'nt', 4352, 900, 900), ('SUE GCN dX', 'nn', 4352, 900, 900), ('SUE inter K', 'nt', 6080, 225, 900),
                             ('SUE bmm A.X', 'nnb', 68, 900, 68), ('tiny', 'nt', 40, 400, 400), ('dW small', 'tn', 400, 200, 3200), ('GCN dW', 'tn', 900, 900, 4352)):
    res = []
    for tile in (2, 6, 0):
        if kind == 'nt':
            a, b, c = torch.randn(M, K, device=d), torch.randn(N, K, device=d), torch.empty(M, N, device=d)
            f = lambda: ops.gemm(a, b, c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=tile)
        elif kind == 'nn':
            a, b, c = torch.randn(M, K, device=d), torch.randn(K, N, device=d), torch.empty(M, N, device=d)
            f = lambda: ops.gemm(a, b, c, M=M, N=N, K=K, lda=K, ldb=N, ldc=N, trans_b=True, tile=tile)
        elif kind == 'nnb':
            a, b, c = torch.randn(64, M, K, device=d), torch.randn(64, K, N, device=d), torch.empty(64, M, N, device=d)
            f = lambda: ops.gemm(a, b, c, M=M, N=N, K=K, lda=K, ldb=N, ldc=N, trans_b=True, batch=64, strideA=M * K, strideB=K * N, strideC=M * N, tile=tile)
        else:
            a, b, c = torch.randn(K, M, device=d), torch.randn(K, N, device=d), torch.zeros(M, N, device=d)
            sk = ops.split_for(M, N, K)
            f = lambda: ops.gemm(a, b, c, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, trans_a=True, trans_b=True, split_k=sk, atomic=True, tile=tile)
        try:
            ms = timeit(f, iters=20)
            res.append('tile %d: %6.1f us' % (tile, ms * 1e3))
        except Exception as e:
            res.append('tile %d: %s' % (tile, str(e)[:30]))
    print('%-12s %-3s M%-5d N%-4d K%-5d | %s' % (label, kind, M, N, K, ' | '.join(res)), flush=True)
