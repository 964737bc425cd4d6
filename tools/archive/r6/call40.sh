#!/bin/bash
# round 6, GPU call 40: why is the driver-line's batch-8 leg (10 steps after 5 warm-up, fresh trainer) 0.4 ms slower than `--batch_size 8 --steps 40 --warmup 8`?
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
B="python bench.py --batch_size 8 --no_cpu_baseline --no_secondary --no_isolated --sustained_seconds 0"
for cfg in "10 5" "10 8" "20 5" "40 8" "10 5"; do
  set -- $cfg
  NNR_BENCH_STEP_MARKS=1 timeout 300 $B --steps $1 --warmup $2 2> gpurun_out/r06L_marks.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $1 warmup $2:', d['ms_per_step'], d['value'])"
  grep -E "step marks" gpurun_out/r06L_marks.txt | cut -c1-200
done
python bench.py --no_cpu_baseline --no_isolated --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['ms_per_step'])
for k,v in d['secondary'].items(): print(k, v['ms_per_step'], v['steps'], v['warmup'])"
