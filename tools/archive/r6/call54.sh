#!/bin/bash
# round 6, GPU call 54: full GPU suite + smoke + the driver's command at the final commit
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06U_tests.log 2>&1; grep -n "passed\|failed" gpurun_out/r06U_tests.log | tail -2
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_final_bench_3.json 2> gpurun_out/r06_final_bench_3.err
python3 - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06_final_bench_3.json') if l.startswith('{')][-1])
print('driver command', d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['roofline']['frac'], (d['roofline'].get('rocprof') or {}).get('frac_in_step'))
for k,v in d['secondary'].items(): print('   ', k, v['ms_per_step'], v['value'])
PY
