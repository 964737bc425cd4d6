#!/bin/bash
# Round 4 A/B: second leaf stream for the title token stream's weight-gradient GEMMs (NNR_LEAF2, default 1)
O=gpurun_out/r04t; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_tape_gpu.py tests/test_hip_model_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 600 python3 -m pytest tests/test_hip_headline_gpu.py -x -q -m gpu -k "identical or config4" >> $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
B="python3 bench.py --prebuilt --no_cpu_baseline --no_isolated --sustained_seconds 2"
for r in 1 2; do
for b in 8 16 64; do
$B --batch_size $b --steps 40 --warmup 8 > $O/bench_b${b}_leaf2_$r.json 2>> $O/err
NNR_LEAF2=0 $B --batch_size $b --steps 40 --warmup 8 > $O/bench_b${b}_one_$r.json 2>> $O/err
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
