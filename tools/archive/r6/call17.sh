#!/bin/bash
# round 6, GPU call 17: the weight-gradient (TN) tile switches of rounds 2-4 re-measured beside the bf16x3 NT kernels (63 KB of LDS, two workgroups per CU):
# the co-residency they were tuned for has changed.  Un-instrumented, two interleaved rounds.
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
rm -f gpurun_out/r06q_ab.txt
ab() {
  echo -n "$1 : " >> gpurun_out/r06q_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06q_ab.txt 2>&1
}
for i in 1 2; do
  ab "X=default"
  ab "NNR_TN_SMALL_LDS=1"
  ab "NNR_TN_WIDE=0"
  ab "NNR_TN_T64=0"
  ab "NNR_TN_T64=1"
  ab "NNR_TN_SQUARE=0"
  ab "NNR_LEAF2=0"
  ab "NNR_POOL_R=16"
done
cat gpurun_out/r06q_ab.txt
