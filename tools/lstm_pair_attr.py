import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops
from tools.lstm_bench import setup, H
crit, _ = setup(16, 128, 43.0, 128)
full, _ = setup(4096, 128, 43.0, 128)
def timeit(fn, items, iters=5):
    fn(items, H); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn(items, H)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
which = ops.lstm_bwd if '--bwd' in sys.argv else ops.lstm_fwd
for dbg, what in ((0, 'full'), (3, 'no exchange at all'),
                  (31, 'barriers + cell update only')):
    os.environ['NNR_LSTM_DBG'] = str(dbg)
    print('dbg %2d %-28s one tile %.2f us/step   4096x128: %.3f ms' % (dbg, what, timeit(which, [crit]) * 1e3 / 128, timeit(which, [full])), flush=True)
