#!/bin/bash
mkdir -p gpurun_out
# round 5: wall time of the driver's default command (headline + sustained + secondary legs + CPU baseline): ~85 s on a fresh box
s=$(date +%s)
python bench.py > gpurun_out/r05x_bench.json 2> gpurun_out/r05x_bench.err
e=$(date +%s)
echo "wall seconds of the driver's command (python bench.py): $((e - s))" | tee gpurun_out/r05x_wall.txt
python -c "import json; d=json.loads(open('gpurun_out/r05x_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], {k: (v.get('ms_per_step'), v.get('leg_seconds')) for k, v in d['secondary'].items()}, d['cpu_baseline']['value'])" | tee -a gpurun_out/r05x_wall.txt
