#!/bin/bash
# round 6, GPU call 36: capacity threshold of the token-stream launches for bf16x3 at batch 64 (title stream: capacity 112 640) and 32
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
rm -f gpurun_out/r06I_ab.txt
ab() {
  echo -n "b$2 $1 : " >> gpurun_out/r06I_ab.txt
  env $1 timeout 300 python bench.py --batch_size $2 --no_cpu_baseline --no_secondary --no_isolated --steps 40 --sustained_seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06I_ab.txt 2>&1
}
for i in 1 2 3 4 5; do
  ab "NNR_BX3_MIN_DYN_ROWS=2048" 64
  ab "NNR_BX3_MIN_DYN_ROWS=120000" 64
  ab "NNR_BX3_MIN_DYN_ROWS=2048" 32
  ab "NNR_BX3_MIN_DYN_ROWS=120000" 32
  ab "NNR_BX3_MIN_DYN_ROWS=250000" 32
done
sort gpurun_out/r06I_ab.txt
