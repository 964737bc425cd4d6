#!/bin/bash
# round 6, GPU call 56: leaf streams confined to a subset of the CUs (hipExtStreamCreateWithCUMask) so that the CU-pair recurrence of the chain always finds free CUs
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
rm -f gpurun_out/r06W_ab.txt
ab() {
  echo -n "$2 | $1 : " >> gpurun_out/r06W_ab.txt
  env $1 timeout 120 python bench.py $2 --no_cpu_baseline --no_secondary --no_isolated --steps 40 --sustained_seconds 1 2>gpurun_out/r06W_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']; print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], {k: v['ms'] for k, v in f.items() if 'lstm' in k})" >> gpurun_out/r06W_ab.txt 2>&1 || { echo FAILED >> gpurun_out/r06W_ab.txt; tail -3 gpurun_out/r06W_err.txt >> gpurun_out/r06W_ab.txt; }
}
ab "NNR_LEAF_CU_MASK=55555555" "--batch_size 64"
cat gpurun_out/r06W_ab.txt
for i in 1 2 3; do
  for m in "" "55555555" "77777777" "0F0F0F0F" "FFFF0000"; do ab "NNR_LEAF_CU_MASK=$m" "--batch_size 64"; done
done
sort gpurun_out/r06W_ab.txt
