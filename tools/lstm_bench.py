#!/usr/bin/env python3
"""Bi-LSTM recurrence micro-benchmark at the history-call shape (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnr_amd import ops
from nnr_amd.layers import LSTMParams
from nnr_amd.synth import _lengths
d = torch.device('cuda')
H, E = 200, 300


def setup(n, Lx, mean, uniform=None):
    rng = np.random.default_rng(0)
    lens = _lengths(rng, n, mean, 0.45, 1, Lx) if uniform is None else np.full(n, uniform)
    mask = torch.from_numpy(np.arange(Lx)[None, :] < lens[:, None]).to(d)
    plan = ops.SeqPlan(mask, None)
    lstm = LSTMParams(E, H).to(d)
    w = ops.LstmPacked(lstm.param_list(), H, E)
    cap = plan.cap
    f = dict(device=d, dtype=torch.float32)
    st = dict(plan=plan, w=w, gates=torch.randn((cap, 2 * w.NP), **f) * 0.5, cell=torch.empty((cap, 2 * w.HP), **f),
              hout=torch.empty((cap, 2 * H), **f), cn=torch.empty((n, 2 * H), **f), dh=torch.randn((cap, 2 * H), **f) * 0.1,
              dcn=torch.randn((n, 2 * H), **f) * 0.1)
    return st, int(lens.sum())


def run(items, tokens, iters=5, label=''):
    for which, fn in (('fwd', ops.lstm_fwd), ('bwd', ops.lstm_bwd)):
        fn(items, H); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn(items, H)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        print('%-40s %s %7.3f ms  %6.1f TF' % (label, which, ms, tokens * 2 * 2.0 * H * 4 * H / ms / 1e9))


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'all'
    c, tc = setup(3200, 128, 43.0)
    t, tt = setup(3200, 32, 11.5)
    if mode == 'crit':
        c3, _ = setup(256, 128, 43.0, uniform=128)
        for _ in range(3):
            ops.lstm_fwd([c3], H); ops.lstm_bwd([c3], H)
        torch.cuda.synchronize(); sys.exit(0)
    if mode == 'attr':
        c3, tc3 = setup(256, 128, 43.0, uniform=128)
        run([c3], tc3, label='crit dbg=%s' % os.environ.get('NNR_LSTM_DBG', '0'))
        sys.exit(0)
    if mode == 'one':
        for _ in range(3):
            ops.lstm_fwd([c, t], H); ops.lstm_bwd([c, t], H)
        torch.cuda.synchronize(); sys.exit(0)
    run([c, t], tc + tt, label='hist call: content+title n=3200')
    run([c], tc, label='content only n=3200')
    c2, tc2 = setup(1600, 128, 43.0)
    run([c2], tc2, label='content only n=1600')
    c3, tc3 = setup(256, 128, 43.0, uniform=128)
    run([c3], tc3, label='n=256 (16 tiles) all len 128')
    c4, tc4 = setup(4096, 128, 43.0, uniform=128)
    run([c4], tc4, label='n=4096 (256 tiles) all len 128')
    c5, tc5 = setup(8192, 128, 43.0, uniform=128)
    run([c5], tc5, label='n=8192 (512 tiles) all len 128')
