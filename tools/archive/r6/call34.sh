#!/bin/bash
# round 6, GPU call 34: small per-GPU batches (the 8-GPU shards): bf16x3 tile / row threshold A/B
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
rm -f gpurun_out/r06G_ab.txt
ab() {
  echo -n "b$2 $1 : " >> gpurun_out/r06G_ab.txt
  env $1 timeout 300 python bench.py --batch_size $2 --no_cpu_baseline --no_secondary --no_isolated --steps 40 --sustained_seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06G_ab.txt 2>&1
}
for i in 1 2 3; do
  for b in 8 16; do
    ab "NNR_X=0" $b
    ab "NNR_BX3_TILE=51" $b
    ab "NNR_BX3_MIN_ROWS=20000" $b
    ab "NNR_BX3_MIN_ROWS=20000 NNR_BX3_TILE=51" $b
    ab "NNR_BX3=0" $b
  done
done
cat gpurun_out/r06G_ab.txt
