#!/usr/bin/env python3
"""Do the DEAD workgroups of capacity-sized launches cost time when two such launches share the chip?  The LSTM input projections of the
batch-64 step: title stream (capacity 112 640 rows, ~20 000 live) and content stream (450 560, ~80 000 live), N = 1664, K = 300, weight
operand (bf16x3 tile unless NNR_BX3=0), issued on two HIP streams as the step does.  Measured: each alone, both together with the grid
sized for the CAPACITY and the live count on the device (`dyn`: what the step launches), both together with the grid sized for the LIVE rows
(static M: no dead workgroup).  Also the gate shape (N = 400, K = 400)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops

d = torch.device('cuda')


def wall(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for N, K in ((1664, 300), (400, 400), (300, 1664)):
    w = torch.nn.Parameter((torch.randn(N, K, device=d) * 0.05))
    specs = {'title': (112640, 20000), 'content': (450560, 80000)}
    bufs = {}
    for k, (cap, live) in specs.items():
        bufs[k] = (torch.randn(cap, K, device=d), torch.empty(cap, N, device=d), torch.tensor([live], device=d, dtype=torch.int32))

    def one(k, dyn):
        a, c, dv = bufs[k]
        cap, live = specs[k]
        if dyn:
            ops.gemm(a, w, c, M=cap, N=N, K=K, lda=K, ldb=K, ldc=N, dyn=dv, dyn_dim=1)
        else:
            ops.gemm(a, w, c, M=live, N=N, K=K, lda=K, ldb=K, ldc=N)

    def both(dyn):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            one('title', dyn)
        with torch.cuda.stream(s2):
            one('content', dyn)
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    fl = {k: 2.0 * specs[k][1] * N * K for k in specs}
    print('N %d K %d (NNR_BX3=%s)' % (N, K, os.environ.get('NNR_BX3', '1')))
    for dyn in (True, False):
        tag = 'capacity grid + dyn' if dyn else 'live grid (static M)'
        t = {k: wall(lambda k=k: one(k, dyn)) for k in specs}
        tb = wall(lambda: both(dyn))
        print('  %-22s title alone %7.1f us %6.1f TF | content alone %7.1f us %6.1f TF | both on two streams %7.1f us %6.1f TF (sum of alone %7.1f)' % (
            tag, 1000 * t['title'], fl['title'] / t['title'] / 1e9, 1000 * t['content'], fl['content'] / t['content'] / 1e9,
            1000 * tb, (fl['title'] + fl['content']) / tb / 1e9, 1000 * (t['title'] + t['content'])))
