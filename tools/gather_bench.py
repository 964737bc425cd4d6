#!/usr/bin/env python3
"""HBM-roofline micro-benchmarks: embedding-row gather / scatter (nn.Embedding fwd / bwd) and the MFMA attention core."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnr_amd import ops
from nnr_amd.synth import _zipf_ids
d = torch.device('cuda')


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


V, E = 60000, 300
rows = 64 * 8800                    # one batch of 64 impressions: 8 800 token slots each
table = torch.randn(V, E, device=d)
rng = np.random.default_rng(0)
for name, ids in (('uniform ids', torch.randint(0, V, (rows,), device=d, dtype=torch.int32)),
                  ('zipf ids', torch.from_numpy(_zipf_ids(rng, rows, 2, V, 1.1)).to(d))):
    out = torch.empty(rows, E, device=d)
    for p in (0.0, 0.2):
        ms = timeit(lambda: ops.embed_gather(table, ids, p, 3, out=out))
        alg = rows * E * 4 * 2            # read the rows once + write them once
        print('embed_gather  %-12s p=%.1f  %7.3f ms  %6.2f TB/s algorithmic (%.0f MB)' % (name, p, ms, alg / ms / 1e9, alg / 1e6))
    g = torch.randn(rows, E, device=d); dt = torch.zeros(V, E, device=d)
    ms = timeit(lambda: ops.embed_scatter(g, ids, dt, 0.2, 3))
    print('embed_scatter %-12s p=0.2  %7.3f ms  %6.2f TB/s of added bytes' % (name, ms, rows * E * 4 / ms / 1e9))
# MFMA attention core at the MHSA news-encoder shape
n, Lq, heads, dh = 3520, 32, 20, 20
HD = heads * dh
qkv = torch.randn(n * Lq, 3 * HD, device=d); mask = (torch.rand(n, Lq, device=d) < 0.4)
mask[:, 0] = True
out = torch.empty(n * Lq, HD, device=d); prob = torch.empty(ops.mhsa_prob_size(n, Lq, heads), device=d)
dout = torch.randn(n * Lq, HD, device=d); dqkv = torch.empty_like(qkv)
ms = timeit(lambda: ops.mhsa_fwd(qkv, mask, n, Lq, heads, dh, out, prob))
fl = n * heads * 4.0 * Lq * Lq * dh
print('mhsa_fwd %d x %d heads, L=%d d=%d  %7.3f ms  %6.2f TFLOP/s algorithmic' % (n, heads, Lq, dh, ms, fl / ms / 1e9))
ms = timeit(lambda: ops.mhsa_bwd(qkv, mask, prob, dout, n, Lq, heads, dh, dqkv))
print('mhsa_bwd                              %7.3f ms  %6.2f TFLOP/s algorithmic' % (ms, 2.5 * fl / ms / 1e9))
if len(sys.argv) > 1:
    torch.cuda.synchronize()
