#!/usr/bin/env python3
"""The attention pools of CNE (csrc/pool.hip) alone, at the in-step shapes of a batch-64 step: 3 520 sequences (3 200 history + 320 candidate
news), D = 400, title stream (L = 32, ~11.5 tokens) and content stream (L = 128, ~43 tokens, padded history slots = 1 token).  The four calls a
token stream makes per step: self pool forward (score = w2 . tanh rows), cross pool forward (dot score), cross pool backward (d score + dv,
no dx), self pool backward with the cross pool's terms folded into the one write of dx.  Prints us per call and GB/s of the algorithmic bytes;
`--check` compares NNR_POOL_TEAM=1 (register-resident rows, round 6) with =0 (one streaming workgroup per sequence) in two child processes.

    python tools/pool_bench.py [--stream content|title] [--dump out.pt]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--stream', default='content')
ap.add_argument('--dump', default=None)
ap.add_argument('--iters', type=int, default=30)
a = ap.parse_args()
from nnr_amd import ops
from nnr_amd.synth import SynthSpec, _lengths

d = torch.device('cuda')
spec = SynthSpec()
rng = np.random.default_rng(3)
n_hist, n_cand = 3200, 320
L, mean = (spec.max_abstract_length, spec.content_len_mean) if a.stream == 'content' else (spec.max_title_length, spec.title_len_mean)
lens = _lengths(rng, n_hist + n_cand, mean, spec.len_sigma, 1, L)
pad = rng.random(n_hist) < 0.5                      # history lengths are uniform on 0..50: half of the 50 slots are the PAD news (1 token)
lens[:n_hist][pad] = 1
lens = torch.from_numpy(np.asarray(lens)).long()
n, D, A = n_hist + n_cand, 400, 200
mask = torch.arange(L)[None, :] < lens[:, None]
plan = ops.SeqPlan(mask.clone().to(d), None)
cap, total = plan.cap, int(lens.sum())
g = torch.Generator().manual_seed(1)
f32 = dict(device=d, dtype=torch.float32)
x = torch.randn(cap, D, generator=g).to(d)
th = torch.tanh(torch.randn(cap, A, generator=g)).to(d)
w2 = (torch.randn(1, A, generator=g) * 0.3).to(d)
v = (torch.randn(n, D, generator=g) * 0.2).to(d)
dout = torch.randn(n, 2 * D, generator=g).to(d)
dself_x = torch.randn(n, D, generator=g).to(d)
alpha_s, alpha_c = torch.zeros(cap, **f32), torch.zeros(cap, **f32)
selfv, rep = torch.empty(n, D, **f32), torch.empty(n, 2 * D, **f32)
ds_c, ds, dv, dHt = torch.zeros(cap, **f32), torch.zeros(cap, **f32), torch.empty(n, D, **f32), torch.zeros(cap, D, **f32)
scale = 1.0 / np.sqrt(A)
kw = dict(x=x, ldx=D, D=D, n=n, Lx=L, plan=plan)
calls = {
    'self_fwd (th rows + x: 2.4 KB/token)': (lambda: ops.pool_fwd(th=th, w2=w2, alpha=alpha_s, out=selfv, ldo=D, **kw), 4.0 * (D + A) + 4),
    'cross_fwd (x once: 1.6 KB/token)': (lambda: ops.pool_fwd(v=v, ldv=D, scale=scale, alpha=alpha_c, out=rep, ldo=2 * D, add_in=selfv, ldadd=D, **kw), 4.0 * D + 4),
    'cross_bwd (x once, no dx)': (lambda: ops.pool_bwd(v=v, ldv=D, scale=scale, alpha=alpha_c, dout=dout, lddo=2 * D, dscore=ds_c, dv=dv, lddv=D, **kw), 4.0 * D + 8),
    'self_bwd (x read + dx written: 3.2 KB/token)': (lambda: ops.pool_bwd(score=None, alpha=alpha_s, dout=dout, lddo=2 * D, dout2=dself_x, lddo2=D, dx=dHt, lddx=D, dscore=ds,
                                                                         alpha_b=alpha_c, dout_b=dout, lddo_b=2 * D, dscore_b=ds_c, v_b=v, ldv_b=D, scale_b=scale, **kw), 8.0 * D + 16),
}
print('%s stream: %d sequences, L %d, %d live tokens (mean %.1f), <= 16 tokens: %d, 17..64: %d, > 64: %d; NNR_POOL_TEAM=%s' % (
    a.stream, n, L, total, total / n, int((lens <= 16).sum()), int(((lens > 16) & (lens <= 64)).sum()), int((lens > 64).sum()), os.environ.get('NNR_POOL_TEAM', '1')))
for name, (fn, per_tok) in calls.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep_ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / a.iters)
    nbytes = per_tok * total + 4.0 * n * D * 3
    print('  %-46s %7.1f us  %6.0f GB/s' % (name, 1000 * best, nbytes / (best * 1e-3) / 1e9))
if a.dump:
    torch.save({k: t.cpu() for k, t in dict(alpha_s=alpha_s, alpha_c=alpha_c, selfv=selfv, rep=rep, ds_c=ds_c, ds=ds, dv=dv, dHt=dHt).items()}, a.dump)
