#!/bin/bash
# round 6, GPU call 33: the driver's command again with the counter files of this build in place and the pools' traffic priced by their packed kernel
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
python3 bench.py > gpurun_out/r06h_bench_again.json 2> gpurun_out/r06h_bench_again.err
python3 - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06h_bench_again.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['sustained']['ms_per_step'])
for k,v in d['roofline']['hbm'].items(): print(k, v['achieved'], v['frac'], v['avg_launch_us'], v.get('traffic_over_algorithmic'))
for k,v in d['secondary'].items(): print(k, v['ms_per_step'], v['value'], v['matrix_path'][:140])
PY
