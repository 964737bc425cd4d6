import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnr_amd import ops
from tools.lstm_bench import setup, H
crit, _ = setup(16, 128, 43.0, 128)
which = ops.lstm_bwd if '--bwd' in sys.argv else ops.lstm_fwd
os.environ['NNR_LSTM_DBG'] = '32'
for _ in range(3):
    which([crit], H)
torch.cuda.synchronize()
sync, _diag_off = ops.LAST_LSTM_SYNC[0]
tb = sync[-(128 * 16 * 2):].view(torch.int64).view(128, 16).cpu().numpy().astype(np.float64)
tb = tb[8:120]
base = tb[:, 4:5]                      # compute wave: step start
names = {4: 'step start', 5: 'partner partials added', 6: 'gate gradients written', 7: 'after barrier', 8: 'partner tile sent', 9: 'own tile done'} if '--bwd' in sys.argv else {2: 'partner tile in LDS', 4: 'compute: step start', 5: 'compute: phase A done',
         6: 'compute: after B2', 7: 'compute: phase B done', 8: 'compute: stores issued', 9: 'compute: activations done', 10: 'compute: tagged stores issued', 11: 'first fetched word returned'}
print('ticks are 10 ns (100 MHz wall clock); mean offset from the compute wave\'s step start, steps 8..119')
for k in sorted(names):
    print('  %-28s %7.2f us' % (names[k], float(np.mean(tb[:, k:k + 1] - base)) / 100.0))
print('  stale words / repoll spins per launch:', int(sync[-(128 * 16 * 2) - 15]), int(sync[-(128 * 16 * 2) - 14]))
print('  step period                  %7.2f us' % (float(np.mean(np.diff(tb[:, 4]))) / 100.0))
