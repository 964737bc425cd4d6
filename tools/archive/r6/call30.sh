#!/bin/bash
# round 6, GPU call 30: configs[1] (MHSA+MHSA) with the bf16x3 kernel per shape class (round 6 measured all classes together: slower)
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
rm -f gpurun_out/r06D_ab.txt
ab() {
  echo -n "$1 : " >> gpurun_out/r06D_ab.txt
  env $1 timeout 300 python bench.py --config mhsa --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06D_ab.txt 2>&1
}
for i in 1 2 3; do
  ab "NNR_BX3_MHSA=0"
  ab "NNR_BX3_MHSA=1 NNR_BX3_CLASSES=dx"
  ab "NNR_BX3_MHSA=1 NNR_BX3_CLASSES=proj"
  ab "NNR_BX3_MHSA=1 NNR_BX3_CLASSES=gate"
  ab "NNR_BX3_MHSA=1 NNR_BX3_CLASSES=dx,proj"
  ab "NNR_BX3_MHSA=1 NNR_BX3_CLASSES=dx,proj,gate"
done
cat gpurun_out/r06D_ab.txt
