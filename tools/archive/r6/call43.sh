#!/bin/bash
# round 6, GPU call 43: GPU_MAX_HW_QUEUES 2 / 3 / 4 / 5 / 6
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
rm -f gpurun_out/r06N_ab.txt
ab() {
  echo -n "$2 | $1 : " >> gpurun_out/r06N_ab.txt
  env $1 timeout 300 python bench.py $2 --no_cpu_baseline --no_secondary --no_isolated --steps 40 --sustained_seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06N_ab.txt 2>&1
}
for i in 1 2; do
  for q in 2 3 4 5 6; do
    ab "GPU_MAX_HW_QUEUES=$q" "--batch_size 64"
    ab "GPU_MAX_HW_QUEUES=$q" "--batch_size 8"
  done
done
sort gpurun_out/r06N_ab.txt
