#!/usr/bin/env python3
"""Weight-gradient (TN) launch shapes of the step under the NNR_TN_WANT knob (minimum workgroup count of the device-side split-K):
run once per value, e.g.  for w in 256 512 768 1024 1536; do NNR_TN_WANT=$w python tools/tn_want_bench.py; done"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops
d = torch.device('cuda')


def t(fn, iters=20):
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


out = []
for name, M, N, live, cap, tile in (('sue dW 900x900x4352', 900, 900, 4352, 4352, 26), ('sue dW 900x900x6080', 900, 900, 6080, 6080, 26),
                                    ('dW_H 400x400 title', 400, 400, 22000, 112640, 26), ('dW_1 200x400 title', 200, 400, 22000, 112640, 26),
                                    ('dW_H 400x400 content', 400, 400, 77000, 450560, 26), ('dW_ih title', 1664, 300, 22000, 112640, 26),
                                    ('dW_ih content', 1664, 300, 77000, 450560, 26)):
    a = torch.randn(cap, M, device=d); b = torch.randn(cap, N, device=d) * 0.05; c = torch.zeros(M, N, device=d)
    dyn = torch.tensor([live], device=d, dtype=torch.int32)
    sk = ops.split_for(M, N, cap, tile_m=128, tile_n=80, target_blocks=4096)
    ms = t(lambda: ops.gemm(a, b, c, M=M, N=N, K=cap, lda=M, ldb=N, ldc=N, trans_a=True, trans_b=True, split_k=sk, atomic=True, tile=tile, dyn=dyn, dyn_dim=2))
    out.append('%s %.1f us %.1f TF' % (name, ms * 1e3, 2.0 * M * N * live / ms / 1e9))
print("WANT=%s STAGES=%s  " % (os.environ.get("NNR_TN_WANT", "default"), os.environ.get("NNR_TN_STAGES", "default")) + ' | '.join(out))
