#!/bin/bash
# Round 4 A/B: title recurrence under the content projection (NNR_LSTM_FWD_SPLIT=1 + NNR_PROJ_ORDER=1)
O=gpurun_out/r04j; mkdir -p $O
B="python3 bench.py --prebuilt --no_cpu_baseline --no_isolated --sustained_seconds 2"
$B > $O/bench_default.json 2>> $O/err
NNR_LSTM_FWD_SPLIT=1 NNR_PROJ_ORDER=1 $B > $O/bench_split_order.json 2>> $O/err
NNR_LSTM_FWD_SPLIT=1 $B > $O/bench_split.json 2>> $O/err
NNR_PROJ_ORDER=1 $B > $O/bench_order.json 2>> $O/err
$B > $O/bench_default2.json 2>> $O/err
NNR_LSTM_FWD_SPLIT=1 NNR_PROJ_ORDER=1 $B > $O/bench_split_order2.json 2>> $O/err
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print('%-28s %8.1f %7.3f sustained %s' % (f.split('bench_')[1][:-5], d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step')))
    except Exception as e: print(f, 'FAILED', e)
PY
tail -5 $O/err
