#!/usr/bin/env python3
"""2-CU pair recurrence vs the one-CU kernel: identical inputs, compare every output; then time both."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops
from tools.lstm_bench import setup, H

def clone(st):
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}

def run(fn, items, pair):
    ops.LSTM_PAIR = pair
    for it in items:
        it.pop('sync', None)
    fn(items, H)
    torch.cuda.synchronize()
    return ops.lstm_sync_timeouts() if pair else 0

for (n, Lx, mean, uni) in ((40, 32, 11.5, None), (333, 128, 43.0, None), (64, 128, 43.0, 128), (3200, 128, 43.0, None)):
    base, tokens = setup(n, Lx, mean, uni)
    a, b = clone(base), clone(base)
    run(ops.lstm_fwd, [a], False)
    to = run(ops.lstm_fwd, [b], True)
    rows = int(a['plan'].total.item())
    errs = {k: float((a[k][:rows] - b[k][:rows]).abs().max()) if k != 'cn' else float((a[k] - b[k]).abs().max()) for k in ('gates', 'cell', 'hout', 'cn')}
    print('fwd n=%d L=%d: timeouts %d, max |pair - one| %s' % (n, Lx, to, errs), flush=True)
    if '--bwd' in sys.argv:
        run(ops.lstm_bwd, [a], False)
        to = run(ops.lstm_bwd, [b], True)
        print('bwd n=%d: timeouts %d, max |d gates| %.3e' % (n, to, float((a['gates'][:rows] - b['gates'][:rows]).abs().max())), flush=True)

def timeit(fn, items, pair, iters=5):
    ops.LSTM_PAIR = pair
    for it in items:
        it.pop('sync', None)
    fn(items, H); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn(items, H)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

c, tc = setup(3200, 128, 43.0)
t, tt = setup(3200, 32, 11.5)
crit, _ = setup(16, 128, 43.0, 128)
full, _ = setup(4096, 128, 43.0, 128)
for label, items in (('hist call content+title n=3200', [c, t]), ('content n=3200', [c]), ('one tile, 128 steps', [crit]), ('4096 x 128 steps', [full])):
    fns = (('fwd', ops.lstm_fwd),) + ((('bwd', ops.lstm_bwd),) if '--bwd' in sys.argv else ())
    for nm, fn in fns:
        print('%-34s %s  one-CU %7.3f ms   pair %7.3f ms' % (label, nm, timeit(fn, items, False), timeit(fn, items, True)), flush=True)
print('timeouts', ops.lstm_sync_timeouts())
