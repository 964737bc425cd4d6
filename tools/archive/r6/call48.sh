#!/bin/bash
# round 6, GPU call 48: the driver's command at the final commit (twice), smoke(), and the stand-alone commands of the legs
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
python3 bench.py > gpurun_out/r06_final_bench_1.json 2> gpurun_out/r06_final_bench_1.err
python3 bench.py > gpurun_out/r06_final_bench_2.json 2> gpurun_out/r06_final_bench_2.err
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for f in 1 2; do python3 - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06_final_bench_$f.json') if l.startswith('{')][-1])
print('run $f', d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['roofline']['frac'], (d['roofline'].get('rocprof') or {}).get('frac_in_step'), d['cpu_baseline']['value'])
for k,v in d['secondary'].items(): print('   ', k, v['ms_per_step'], v['value'], v['matrix_path'][:30])
PY
done
