#!/bin/bash
# Round 4 A/B: SUE's candidate-side projections / candidate-gradient accumulations on a side stream (NNR_SUE_SIDE, default 1)
O=gpurun_out/r04l; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_tape_gpu.py tests/test_hip_layers_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="python3 bench.py --prebuilt --no_cpu_baseline --no_isolated --sustained_seconds 2"
for r in 1 2; do
$B > $O/bench_side_$r.json 2>> $O/err
NNR_SUE_SIDE=0 $B > $O/bench_noside_$r.json 2>> $O/err
$B --batch_size 8 --steps 40 --warmup 8 > $O/bench_b8_side_$r.json 2>> $O/err
NNR_SUE_SIDE=0 $B --batch_size 8 --steps 40 --warmup 8 > $O/bench_b8_noside_$r.json 2>> $O/err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print('%-20s %8.1f %7.3f sustained %s' % (f.split('bench_')[1][:-5], d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step')))
    except Exception as e: print(f, 'FAILED', e)
PY
tail -3 $O/err
