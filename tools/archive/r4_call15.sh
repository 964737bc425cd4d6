#!/bin/bash
# Round 4: sorted scatter with 8 rows in flight (solo figures + step), unit tests of the scatter / determinism
O=gpurun_out/r04m; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_ops_gpu.py -x -q -m gpu -k "scatter or sorted or embed" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python3 -m pytest tests/test_hip_headline_gpu.py -x -q -m gpu -k "identical" >> $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
python3 tools/scatter_bench.py > $O/scatter_bench.txt 2>&1; cat $O/scatter_bench.txt
python3 tools/scatter_bench.py --batch_size 8 > $O/scatter_bench_b8.txt 2>&1; cat $O/scatter_bench_b8.txt
B="python3 bench.py --prebuilt --no_cpu_baseline --no_isolated --sustained_seconds 2"
for r in 1 2; do
$B > $O/bench_default_$r.json 2>> $O/err
$B --batch_size 8 --steps 40 --warmup 8 > $O/bench_b8_$r.json 2>> $O/err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print('%-20s %8.1f %7.3f sustained %s' % (f.split('bench_')[1][:-5], d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step')))
    except Exception as e: print(f, 'FAILED', e)
PY
