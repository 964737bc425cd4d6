#!/bin/bash
# Round 4 A/B: quad-tile thresholds after the balanced backward recurrence (defaults at batch 64: 96 / 96)
O=gpurun_out/r04k; mkdir -p $O
B="python3 bench.py --prebuilt --no_cpu_baseline --no_isolated --sustained_seconds 2"
$B > $O/bench_default.json 2>> $O/err
NNR_LSTM_QUAD_T=80 $B > $O/bench_t80.json 2>> $O/err
NNR_LSTM_QUAD_T=112 $B > $O/bench_t112.json 2>> $O/err
NNR_LSTM_QUAD_T=96 NNR_LSTM_QUAD_T_BWD=80 $B > $O/bench_bwd80.json 2>> $O/err
NNR_LSTM_QUAD_T=96 NNR_LSTM_QUAD_T_BWD=112 $B > $O/bench_bwd112.json 2>> $O/err
NNR_LSTM_QUAD_T=96 NNR_LSTM_QUAD_T_BWD=128 $B > $O/bench_bwd128.json 2>> $O/err
NNR_LSTM_QUAD_T=80 NNR_LSTM_QUAD_T_BWD=112 $B > $O/bench_t80_bwd112.json 2>> $O/err
$B > $O/bench_default2.json 2>> $O/err
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); fam=d['roofline']['families']
        print('%-20s %8.1f %7.3f sustained %s  lstm_fwd %.3f ms lstm_bwd %.3f ms (sampled)' % (f.split('bench_')[1][:-5], d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step'), fam['lstm_fwd']['ms'], fam['lstm_bwd']['ms']))
    except Exception as e: print(f, 'FAILED', e)
PY
