#!/bin/bash
# round 5, GPU call 7: EXPERIMENTAL bf16x3 NT GEMMs inside the library (NNR_BX3=1, off by default): unit + headline parity, in-step A/B
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_hip_ops_gpu.py tests/test_hip_headline_gpu.py -q -x -k "bf16x3" -s 2>&1 | grep -v amdgpu.ids | tail -12) > gpurun_out/r05i_tests.log
tail -6 gpurun_out/r05i_tests.log
ab() {
  echo "$1 $2" >> gpurun_out/r05i_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], {k: (v['ms'], v['tflops'], v['launches']) for k, v in d['roofline']['families'].items() if 'nt_' in k})" >> gpurun_out/r05i_ab.txt
}
for i in 1 2; do
  ab "NNR_BX3=0" ""
  ab "NNR_BX3=1" ""
done
ab "NNR_BX3=0" "--batch_size 8"
ab "NNR_BX3=1" "--batch_size 8"
ab "NNR_BX3=0" "--batch_size 128"
ab "NNR_BX3=1" "--batch_size 128"
ab "NNR_BX3=0" "--config mhsa"
ab "NNR_BX3=1" "--config mhsa"
cat gpurun_out/r05i_ab.txt
