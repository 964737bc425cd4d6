#!/bin/bash
# round 6, GPU call 37: size rule of the bf16x3 path (NNR_BX3_MIN_SEQS=1408: batch 8 / 16 on the fp32 kernels): GPU suite + batch sweep
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06J_tests.log 2>&1
tail -3 gpurun_out/r06J_tests.log
rm -f gpurun_out/r06J_ab.txt
ab() {
  echo -n "b$2 $1 : " >> gpurun_out/r06J_ab.txt
  env $1 timeout 300 python bench.py --batch_size $2 --no_cpu_baseline --no_secondary --no_isolated --steps 40 --sustained_seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], d['config']['matrix_path']['nt_weight_gemms'][:40])" >> gpurun_out/r06J_ab.txt 2>&1
}
for i in 1 2; do for b in 8 16 32 64; do ab "NNR_BX3_MIN_SEQS=0" $b; ab "NNR_BX3_MIN_SEQS=1408" $b; done; done
cat gpurun_out/r06J_ab.txt
