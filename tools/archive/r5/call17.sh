#!/bin/bash
# round 5, GPU call 17: fixed-order stream-K NT kernel (tile 47): unit test, alone vs the 128 x 80 families, in-step A/B (NNR_SK=1)
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q --tb=short -k "stream_k" 2>&1 | grep -v amdgpu.ids | tail -25) > gpurun_out/r05s_tests.log
tail -25 gpurun_out/r05s_tests.log
SHAPES=sue TILES=9,47 ROUNDS=5 timeout 300 python tools/gemm_pipe_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05s_sue_tiles.txt
if [ "$1" = "step" ]; then
  rm -f gpurun_out/r05s_ab.txt
  ab() {
    echo "$1 $2" >> gpurun_out/r05s_ab.txt
    env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], {k: (v['ms'], v['tflops'], v['launches']) for k, v in d['roofline']['families'].items() if 'sk_' in k or 'pipe2_128x80' in k})" >> gpurun_out/r05s_ab.txt 2>&1
  }
  for i in 1 2 3; do ab "NNR_SK=0" ""; ab "NNR_SK=1" ""; done
  ab "NNR_SK=0" "--batch_size 32"; ab "NNR_SK=1" "--batch_size 32"
  cat gpurun_out/r05s_ab.txt
fi
