#!/bin/bash
# round 6, GPU call 41: the batch-8 leg now draws the batches a stand-alone `--batch_size 8 --steps 10 --warmup 5` run draws: leg vs stand-alone, one box
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
B="python bench.py --batch_size 8 --no_cpu_baseline --no_secondary --no_isolated --sustained_seconds 0 --steps 10 --warmup 5"
for i in 1 2; do timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stand-alone b8 10/5:', d['ms_per_step'], d['value'])"; done
timeout 300 python bench.py --config mhsa --no_cpu_baseline --no_secondary --no_isolated --sustained_seconds 0 --steps 10 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stand-alone mhsa 10/5:', d['ms_per_step'], d['value'])"
for i in 1 2; do python bench.py --no_cpu_baseline --no_isolated --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['ms_per_step'])
for k,v in d['secondary'].items(): print('  leg', k, v['ms_per_step'], v['steps'], v['warmup'])"; done
