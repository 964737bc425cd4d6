#!/bin/bash
# round 5, GPU call 2: gen-3 persistent NT kernel (tiles 40-42): unit tests, isolated rates, in-step A/B
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_hip_ops_gpu.py -q -x -k "persistent_nt" 2>&1 | tail -15) > gpurun_out/r05d_tests.log
tail -5 gpurun_out/r05d_tests.log
TILES=15,9,40,41,42 ROUNDS=3 timeout 600 python tools/gemm_pipe_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05d_gemm_bench.txt
ab() {
  echo "$1" >> gpurun_out/r05d_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], {k: (v['ms'], v['tflops']) for k, v in d['roofline']['families'].items() if 'nt_pipe' in k})" >> gpurun_out/r05d_ab.txt
}
for i in 1 2; do
  ab "NNR_X=0"
  ab "NNR_GATE_TILE=40"
  ab "NNR_PROJ_TILE=40"
  ab "NNR_GATE_TILE=40 NNR_PROJ_TILE=40"
  ab "NNR_GATE_TILE=40 NNR_PROJ_TILE=40 NNR_DX_TILE=40 NNR_TITLE_DX_TILE=40"
done
cat gpurun_out/r05d_ab.txt
