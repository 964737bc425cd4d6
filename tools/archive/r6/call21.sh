#!/bin/bash
# round 6, GPU call 21: bf16x3 weight-image prefetch moved in FRONT of the token sorts (behind the LSTM weight packing, under the input projection): tape tests + A/B
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
(timeout 900 python -m pytest tests/test_hip_tape_gpu.py tests/test_hip_headline_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -v amdgpu.ids | tail -8) > gpurun_out/r06u_tests.log
tail -3 gpurun_out/r06u_tests.log | cut -c1-200
rm -f gpurun_out/r06u_ab.txt
ab() {
  echo -n "$1 $2: " >> gpurun_out/r06u_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06u_ab.txt 2>&1
}
for i in 1 2 3; do ab "NNR_BX3_PREFETCH=0" ""; ab "NNR_BX3_PREFETCH=1" ""; done
ab "NNR_BX3_PREFETCH=0" "--batch_size 8"; ab "NNR_BX3_PREFETCH=1" "--batch_size 8"
cat gpurun_out/r06u_ab.txt
timeout 300 python tools/tape_timeline.py --batch_size 64 2>&1 | grep -v amdgpu.ids | head -40 > gpurun_out/r06u_timeline_head.txt
