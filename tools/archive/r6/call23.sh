#!/bin/bash
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
rm -f gpurun_out/r06w_ab.txt
ab() {
  echo -n "$1 : " >> gpurun_out/r06w_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06w_ab.txt 2>&1
}
for i in 1 2; do
  ab "X=default"
  ab "NNR_TN_WANT=256"
  ab "NNR_TN_WANT=1024"
  ab "NNR_TN_STAGES=48"
  ab "NNR_TN_STAGES=192"
  ab "NNR_TN_WANT=256 NNR_TN_STAGES=192"
done
cat gpurun_out/r06w_ab.txt
