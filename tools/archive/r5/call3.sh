#!/bin/bash
# round 5, GPU call 3: gen-3 NT kernel, persistent grid size A/B (NNR_P3_SLOTS: 512 = persistent, 100000000 = one tile per workgroup)
mkdir -p gpurun_out
for sl in 100000000 1024 256; do
  echo "NNR_P3_SLOTS=$sl" | tee -a gpurun_out/r05e_gemm_bench.txt
  NNR_P3_SLOTS=$sl TILES=15,9,40 ROUNDS=3 timeout 600 python tools/gemm_pipe_bench.py 2>&1 | grep -v amdgpu.ids | head -5 | tee -a gpurun_out/r05e_gemm_bench.txt
done
ab() {
  echo "$1" >> gpurun_out/r05e_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], {k: (v['ms'], v['tflops']) for k, v in d['roofline']['families'].items() if 'nt_pipe' in k})" >> gpurun_out/r05e_ab.txt
}
for i in 1 2; do
  ab "NNR_X=0"
  ab "NNR_GATE_TILE=40 NNR_P3_SLOTS=100000000"
  ab "NNR_GATE_TILE=40 NNR_PROJ_TILE=40 NNR_P3_SLOTS=100000000"
  ab "NNR_GATE_TILE=40 NNR_P3_SLOTS=1024"
done
cat gpurun_out/r05e_ab.txt
