#!/bin/bash
# round 6, GPU call 50: GPU_MAX_HW_QUEUES pinned by the package (= HIP's default): headline / batch 8 / MHSA unchanged; the tape + DP tests
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
for c in "--batch_size 64" "--batch_size 8" "--config mhsa"; do
  timeout 300 python bench.py $c --no_cpu_baseline --no_secondary --no_isolated --steps 40 --sustained_seconds 1 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])"
done
timeout 1500 python -m pytest tests -x -q -m gpu -k "tape or dp or headline" > gpurun_out/r06S_tests.log 2>&1; tail -2 gpurun_out/r06S_tests.log
