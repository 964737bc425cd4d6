#!/bin/bash
# round 5, GPU call 1: fused pool backward (unit + parity tests, same-box A/B), bf16x3 micro-benchmark
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_hip_ops_gpu.py tests/test_hip_headline_gpu.py tests/test_hip_model_gpu.py tests/test_hip_dropout_gpu.py -q -x -k "pool or kink or REPLAYED_step or golden or fixture or dropout" 2>&1 | tail -15) > gpurun_out/r05c_tests.log
tail -3 gpurun_out/r05c_tests.log
for shape in "450560 400 400" "450560 200 400" "450560 400 200" "4352 900 904"; do
  timeout 300 tools/micro/bf16x3_gemm $shape 2>&1 | tee -a gpurun_out/r05c_bf16x3.txt
done
for i in 1 2; do
  for v in 0 1; do
    echo "NNR_POOL_FUSED=$v round $i" >> gpurun_out/r05c_ab.txt
    NNR_POOL_FUSED=$v timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['roofline']['hbm'].items() if k.startswith('pool')})" >> gpurun_out/r05c_ab.txt
  done
done
cat gpurun_out/r05c_ab.txt
