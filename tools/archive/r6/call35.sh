#!/bin/bash
# round 6, GPU call 35: capacity threshold of the token-stream launches for the bf16x3 kernel, per-GPU batch 8 / 16 / 32 / 64
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
rm -f gpurun_out/r06H_ab.txt
ab() {
  echo -n "b$2 $1 : " >> gpurun_out/r06H_ab.txt
  env $1 timeout 300 python bench.py --batch_size $2 --no_cpu_baseline --no_secondary --no_isolated --steps 40 --sustained_seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06H_ab.txt 2>&1
}
for i in 1 2; do
  for b in 8 16 32 64; do
    ab "NNR_BX3_MIN_DYN_ROWS=2048" $b
    ab "NNR_BX3_MIN_DYN_ROWS=60000" $b
    ab "NNR_BX3_MIN_DYN_ROWS=120000" $b
    ab "NNR_BX3_MIN_DYN_ROWS=250000" $b
    ab "NNR_BX3_MIN_DYN_ROWS=1000000" $b
    ab "NNR_BX3=0" $b
  done
done
cat gpurun_out/r06H_ab.txt
