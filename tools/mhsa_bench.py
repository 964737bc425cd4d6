"""Time the attention core alone (BASELINE configs[1] shape: 64*55 titles of 32 tokens, 20 heads x 20): fwd and bwd, with
the probabilities saved or recomputed.  Usage: python tools/mhsa_bench.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnr_amd import ops

d = torch.device('cuda:0')
n, Lq, heads, dh = 64 * 55, 32, 20, 20
HD = heads * dh
qkv = torch.randn(n * Lq, 3 * HD, device=d)
mask = torch.rand(n, Lq, device=d) < 0.6
out = torch.empty(n * Lq, HD, device=d)
dout = torch.randn(n * Lq, HD, device=d)
dqkv = torch.empty_like(qkv)
prob = torch.empty(ops.mhsa_prob_size(n, Lq, heads), device=d)


def t(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1000


io_f = (qkv.numel() + out.numel()) * 4
io_b = (2 * qkv.numel() + out.numel()) * 4
for name, p in (('saved', prob), ('recompute', None)):
    f = t(lambda: ops.mhsa_fwd(qkv, mask, n, Lq, heads, dh, out, p))
    b = t(lambda: ops.mhsa_bwd(qkv, mask, p, dout, n, Lq, heads, dh, dqkv))
    extra = prob.numel() * 4 if p is not None else 0
    print(f'{name:10s} fwd {f:7.1f} us ({(io_f + extra) / f / 1e3:6.0f} GB/s)   bwd {b:7.1f} us ({(io_b + extra) / b / 1e3:6.0f} GB/s)')
