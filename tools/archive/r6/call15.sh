#!/bin/bash
# round 6, GPU call 15 (= call 14 again): grouped short titles (pairs + quads) in the MHSA attention core (unit + model / tape tests, in-step A/B), bf16x3 image prefetch A/B
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
(timeout 1500 python -m pytest tests/test_hip_ops_gpu.py tests/test_hip_tape_gpu.py tests/test_hip_model_gpu.py tests/test_hip_layers_gpu.py tests/test_hip_headline_gpu.py -m gpu -q --tb=short -k "mhsa or MHSA" 2>&1 | grep -v amdgpu.ids | tail -25) > gpurun_out/r06o_tests.log
tail -12 gpurun_out/r06o_tests.log | cut -c1-300
rm -f gpurun_out/r06o_ab.txt
ab() {
  echo -n "$1 $2: " >> gpurun_out/r06o_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; m=r.get('mhsa') or {}; print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], 'dominant', r['family'], r['avg_launch_us'], 'mhsa', {k: v['avg_launch_us'] for k, v in m.items()})" >> gpurun_out/r06o_ab.txt 2>&1
}
for i in 1 2 3; do ab "NNR_MHSA_PAIR=0" "--config mhsa"; ab "NNR_MHSA_PAIR=1" "--config mhsa"; done
for i in 1 2 3; do ab "NNR_BX3_PREFETCH=0" ""; ab "NNR_BX3_PREFETCH=1" ""; done
cat gpurun_out/r06o_ab.txt
