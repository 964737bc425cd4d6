#!/bin/bash
# round 5, GPU call 8: bf16x3 cache-identity fix (suite under NNR_BX3=1), tile variants 50-53 in the step
mkdir -p gpurun_out
(NNR_BX3=1 timeout 1500 python -m pytest tests -m gpu -q --tb=line 2>&1 | grep -v amdgpu.ids | tail -15) > gpurun_out/r05j_suite_bx3.log
tail -5 gpurun_out/r05j_suite_bx3.log
ab() {
  echo "$1 $2" >> gpurun_out/r05j_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], {k: (v['ms'], v['tflops'], v['launches']) for k, v in d['roofline']['families'].items() if 'bx3' in k})" >> gpurun_out/r05j_ab.txt
}
for i in 1 2; do
  ab "NNR_BX3=1 NNR_BX3_TILE=50" ""
  ab "NNR_BX3=1 NNR_BX3_TILE=51" ""
  ab "NNR_BX3=1 NNR_BX3_TILE=52" ""
  ab "NNR_BX3=1 NNR_BX3_TILE=53" ""
done
ab "NNR_BX3=1 NNR_BX3_TILE=50 NNR_BX3_MIN_ROWS=8192" ""
cat gpurun_out/r05j_ab.txt
