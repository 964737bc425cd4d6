#!/bin/bash
# round 6, GPU call 49: ops.STEP_ROWS was stale in the CNE+SUE step (left by the previous leg's MHSA step): legs vs stand-alone with the step posting 0; and Model.forward's rule as an A/B
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
sa() { env $2 timeout 300 python bench.py $1 --no_cpu_baseline --no_secondary --no_isolated --sustained_seconds 0 --steps 10 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stand-alone $1 $2 10/5:', d['ms_per_step'], d['value'])"; }
sa "--batch_size 8" "NNR_X=0"; sa "--batch_size 8" "NNR_CNE_STEP_ROWS=1"; sa "--batch_size 16 --vocabulary_size 130000" "NNR_X=0"; sa "--batch_size 16 --vocabulary_size 130000" "NNR_CNE_STEP_ROWS=1"; sa "--config mhsa" "NNR_X=0"
for i in 1 2 3; do for e in "NNR_X=0" "NNR_CNE_STEP_ROWS=1"; do env $e timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 --sustained_seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b64 $e', d['ms_per_step'], d['sustained']['ms_per_step'])"; done; done
for i in 1 2; do python bench.py --no_cpu_baseline --no_isolated --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['ms_per_step'], d['sustained']['ms_per_step'])
for k,v in d['secondary'].items(): print('  leg', k, v['ms_per_step'])"; done
