#!/bin/bash
# Round 4 A/B: order of the content stream's tail on the main stream (NNR_DWHH_FIRST=1: dW_hh(forward) before dX + scatter)
O=gpurun_out/r04n; mkdir -p $O
B="python3 bench.py --prebuilt --no_cpu_baseline --no_isolated --sustained_seconds 2"
for r in 1 2; do
$B > $O/bench_default_$r.json 2>> $O/err
NNR_DWHH_FIRST=1 $B > $O/bench_dwhh_first_$r.json 2>> $O/err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print('%-20s %8.1f %7.3f sustained %s' % (f.split('bench_')[1][:-5], d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step')))
    except Exception as e: print(f, 'FAILED', e)
PY
