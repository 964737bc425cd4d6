#!/bin/bash
# Round 4 A/B: user encoder's leaf-stream join deferred to the end of the step (NNR_SUE_JOIN=0) at batch 8 / 16 / 64
O=gpurun_out/r04r; mkdir -p $O
B="python3 bench.py --prebuilt --no_cpu_baseline --no_isolated --sustained_seconds 2"
for r in 1 2; do
for b in 8 16 64; do
$B --batch_size $b --steps 40 --warmup 8 > $O/bench_b${b}_join_$r.json 2>> $O/err
NNR_SUE_JOIN=0 $B --batch_size $b --steps 40 --warmup 8 > $O/bench_b${b}_defer_$r.json 2>> $O/err
done
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print('%-16s %8.1f %7.3f sustained %s' % (f.split('bench_')[1][:-5], d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step')))
    except Exception as e: print(f, 'FAILED', e)
PY
