#!/bin/bash
# round 5, GPU call 9: fused one-launch GCN layers for small batches (default suite + A/B at batch 8 / 16), bf16x3 suite after the cache-identity fix, bx3 tile variants
mkdir -p gpurun_out
rm -f gpurun_out/r05k_ab.txt
(timeout 1500 python -m pytest tests -m gpu -q --tb=line 2>&1 | grep -v amdgpu.ids | tail -15) > gpurun_out/r05k_suite.log
tail -4 gpurun_out/r05k_suite.log
ab() {
  echo "$1 $2" >> gpurun_out/r05k_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], {k: (v['ms'], v['tflops'], v['launches']) for k, v in d['roofline']['families'].items() if 'bx3' in k})" >> gpurun_out/r05k_ab.txt 2>&1
}
for i in 1 2; do
  ab "NNR_GCN_SMALL_ROWS=0" "--batch_size 8"
  ab "NNR_GCN_SMALL_ROWS=1100" "--batch_size 8"
  ab "NNR_GCN_SMALL_ROWS=0" "--batch_size 16"
  ab "NNR_GCN_SMALL_ROWS=1100" "--batch_size 16"
done
ab "NNR_GCN_SMALL_ROWS=2200" "--batch_size 32"
ab "NNR_GCN_SMALL_ROWS=0" "--batch_size 32"
cat gpurun_out/r05k_ab.txt
(NNR_BX3=1 timeout 1500 python -m pytest tests -m gpu -q --tb=line 2>&1 | grep -v amdgpu.ids | tail -15) > gpurun_out/r05k_suite_bx3.log
tail -4 gpurun_out/r05k_suite_bx3.log
for i in 1 2; do
  ab "NNR_BX3=1 NNR_BX3_TILE=50" ""
  ab "NNR_BX3=1 NNR_BX3_TILE=51" ""
  ab "NNR_BX3=1 NNR_BX3_TILE=52" ""
  ab "NNR_BX3=1 NNR_BX3_TILE=53" ""
done
ab "NNR_BX3=1 NNR_BX3_TILE=50 NNR_BX3_MIN_ROWS=8192" ""
tail -12 gpurun_out/r05k_ab.txt
