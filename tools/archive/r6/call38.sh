#!/bin/bash
# round 6, GPU call 38: is the batch-8 step host-bound?  host enqueue timestamps vs GPU step marks, bf16x3 on / off
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
for v in 0 1408; do
  NNR_BX3_MIN_SEQS=$v NNR_BENCH_STEP_MARKS=1 timeout 300 python bench.py --batch_size 8 --no_cpu_baseline --no_secondary --no_isolated --steps 20 --sustained_seconds 0 2> gpurun_out/r06K_marks_$v.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MIN_SEQS=$v', d['ms_per_step'], d['value'])"
  grep -E "step marks|host enqueue" gpurun_out/r06K_marks_$v.txt
done
nproc; grep -m1 "model name" /proc/cpuinfo
