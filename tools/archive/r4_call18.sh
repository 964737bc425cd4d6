#!/bin/bash
# Round 4: persistent MHSA backward (NNR_MHSA_PERSIST, default 1): unit test, solo timing, MHSA+MHSA step
O=gpurun_out/r04o; mkdir -p $O
timeout 600 python3 -m pytest tests/test_hip_ops_gpu.py -x -q -m gpu -k "mhsa" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
echo "persist=1"; python3 tools/mhsa_bench.py 2>&1 | grep -v amdgpu
echo "persist=0"; NNR_MHSA_PERSIST=0 python3 tools/mhsa_bench.py 2>&1 | grep -v amdgpu
B="python3 bench.py --config mhsa --no_cpu_baseline --no_isolated --sustained_seconds 2"
for r in 1 2; do
$B > $O/bench_persist_$r.json 2>> $O/err
NNR_MHSA_PERSIST=0 $B > $O/bench_nopersist_$r.json 2>> $O/err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); m=d['roofline']['mhsa']['mhsa_bwd']
        print('%-20s %8.1f %7.3f sustained %s  mhsa_bwd %.0f us in-step' % (f.split('bench_')[1][:-5], d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step'), m['avg_launch_us']))
    except Exception as e: print(f, 'FAILED', e)
PY
