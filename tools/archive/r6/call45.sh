#!/bin/bash
# round 6, GPU call 45: stream sets aligned to the hardware-queue round-robin: the driver line's legs against their stand-alone commands, one box
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
sa() { timeout 300 python bench.py $1 --no_cpu_baseline --no_secondary --no_isolated --sustained_seconds 0 --steps 10 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stand-alone $1 10/5:', d['ms_per_step'], d['value'])"; }
sa "--batch_size 8"; sa "--batch_size 8"; sa "--config mhsa"; sa "--batch_size 16 --vocabulary_size 130000"
NNR_STREAM_BURN=4 timeout 300 python bench.py --batch_size 8 --no_cpu_baseline --no_secondary --no_isolated --sustained_seconds 0 --steps 10 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stand-alone b8, 4 idle streams in front:', d['ms_per_step'])"
for i in 1 2; do python bench.py --no_cpu_baseline --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['ms_per_step'], d['sustained']['ms_per_step'])
for k,v in d['secondary'].items(): print('  leg', k, v['ms_per_step'], v['steps'], v['warmup'])"; done
timeout 900 python -m pytest tests -x -q -m gpu -k "tape or headline or replay" > gpurun_out/r06P_tests.log 2>&1; tail -2 gpurun_out/r06P_tests.log
