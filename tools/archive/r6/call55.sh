#!/bin/bash
# round 6, GPU call 55: fewer HIP streams at batch 64 / 32 (5 streams on 4 hardware queues): second leaf stream off, SUE side stream off
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
rm -f gpurun_out/r06V_ab.txt
ab() {
  echo -n "$2 | $1 : " >> gpurun_out/r06V_ab.txt
  env $1 timeout 300 python bench.py $2 --no_cpu_baseline --no_secondary --no_isolated --steps 40 --sustained_seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06V_ab.txt 2>&1
}
for i in 1 2 3 4; do
  for e in "NNR_X=0" "NNR_LEAF2=0" "NNR_SUE_SIDE=0" "NNR_LEAF2=0 NNR_SUE_SIDE=0"; do
    ab "$e" "--batch_size 64"
  done
done
for i in 1 2; do for e in "NNR_X=0" "NNR_LEAF2=0"; do ab "$e" "--batch_size 32"; done; done
sort gpurun_out/r06V_ab.txt
