#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r05t_var.txt
for v in 0 4; do
  echo "NNR_SK_VAR=$v" >> gpurun_out/r05t_var.txt
  (NNR_SK_VAR=$v timeout 300 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q --tb=line -k "stream_k" 2>&1 | grep -v amdgpu.ids | tail -5) >> gpurun_out/r05t_var.txt
  NNR_SK_VAR=$v SHAPES=sue TILES=9,47 ROUNDS=5 timeout 300 python tools/gemm_pipe_bench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05t_var.txt
done
cat gpurun_out/r05t_var.txt
