#!/bin/bash
# round 6, GPU call 57: HIP runtime knobs for launch latency: HIP_FORCE_DEV_KERNARG (kernel arguments in device memory), ROC_SIGNAL_POOL / active wait
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
rm -f gpurun_out/r06X_ab.txt
ab() {
  echo -n "$2 | $1 : " >> gpurun_out/r06X_ab.txt
  env $1 timeout 200 python bench.py $2 --no_cpu_baseline --no_secondary --no_isolated --steps 40 --sustained_seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06X_ab.txt 2>&1 || echo FAILED >> gpurun_out/r06X_ab.txt
}
for i in 1 2 3; do
  for e in "NNR_X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; do
    ab "$e" "--batch_size 64"; ab "$e" "--batch_size 8"; ab "$e" "--config mhsa"
  done
done
sort gpurun_out/r06X_ab.txt
