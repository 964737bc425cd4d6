#!/bin/bash
# round 5, GPU call 22: ragged last column block / reduction tail skipped in the NT kernels (NNR_RAGGED A/B): GEMM tests, alone, in the step
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_hip_ops_gpu.py tests/test_hip_layers_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -v amdgpu.ids | tail -8) > gpurun_out/r05u_tests.log
tail -4 gpurun_out/r05u_tests.log
rm -f gpurun_out/r05u_alone.txt
for r in 0 1; do
  echo "NNR_RAGGED=$r" >> gpurun_out/r05u_alone.txt
  NNR_RAGGED=$r TILES=9,15 ROUNDS=5 timeout 300 python tools/gemm_pipe_bench.py 2>&1 | grep -v amdgpu.ids | grep -E "sue|att |dX|gate" >> gpurun_out/r05u_alone.txt
done
cat gpurun_out/r05u_alone.txt
rm -f gpurun_out/r05u_ab.txt
ab() {
  echo "$1 $2" >> gpurun_out/r05u_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r05u_ab.txt 2>&1
}
for i in 1 2 3; do ab "NNR_RAGGED=0" ""; ab "NNR_RAGGED=1" ""; done
ab "NNR_RAGGED=0" "--batch_size 8"; ab "NNR_RAGGED=1" "--batch_size 8"
ab "NNR_RAGGED=0" "--config mhsa"; ab "NNR_RAGGED=1" "--config mhsa"
cat gpurun_out/r05u_ab.txt
