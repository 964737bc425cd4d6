#!/usr/bin/env python3
"""The mid-size NT GEMMs of the MHSA user encoder (3 200 history rows; configs[1]) alone on the GPU, per tile: the automatic choice (0) against
the register-staged tiles (2, 4, 5, 6), the K-splitting skinny kernel (7), the pipelined tiles (15, 16) and the bf16x3 tiles (50, 51: B marked as a weight)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops
from tools.gemm_bench import timeit
d = torch.device('cuda')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
for label, N, K in (('QKV', 1200, 500), ('out proj', 500, 400), ('att affine', 200, 500), ('d att', 500, 200), ('d out proj', 400, 500), ('d QKV', 500, 1200)):
    a, b, c = torch.randn(M, K, device=d), torch.randn(N, K, device=d), torch.empty(M, N, device=d)
    ops.mark_weight(b)
    res = []
    for tile in (0, 2, 4, 5, 6, 7, 15, 16, 50, 51):
        prev = ops.BX3[0]
        ops.BX3[0] = tile in (50, 51)
        kw = {} if tile in (50, 51) else {'tile': tile}
        if tile in (50, 51):
            os.environ['NNR_BX3_TILE'] = str(tile)
            ops._BX3_TILE = tile
            ops._BX3_CLASSES = set(['dx', 'sue', 'proj', 'gate', 'other', ops.bx3_class(N, K)])
        f = lambda: ops.gemm(a, b, c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **kw)
        try:
            ms = timeit(f, iters=30)
            res.append('%d: %5.1f' % (tile, ms * 1e3))
        except Exception as e:
            res.append('%d: %s' % (tile, str(e)[:20]))
        ops.BX3[0] = prev
    print('%-10s M%-5d N%-4d K%-5d us | %s' % (label, M, N, K, ' | '.join(res)), flush=True)
