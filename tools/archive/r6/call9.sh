#!/bin/bash
# round 6, GPU call 9: corrected truncation split (scalar form) + packed pools R = 8 by default: GEMM / pool unit tests, headline + model parity, in-step A/B
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
(timeout 1200 python -m pytest tests/test_hip_ops_gpu.py tests/test_hip_headline_gpu.py tests/test_hip_model_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -v amdgpu.ids | tail -15) > gpurun_out/r06i_tests.log
tail -6 gpurun_out/r06i_tests.log | cut -c1-250
NNR_BX3=1 timeout 300 python tools/dyn_pair_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06i_dyn_pair.txt
rm -f gpurun_out/r06i_ab.txt
ab() {
  echo -n "$1 $2: " >> gpurun_out/r06i_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['roofline']['hbm']; print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], 'pool_bwd', h['pool_bwd']['avg_launch_us'], 'pool_fwd', h['pool_fwd']['avg_launch_us'], 'dominant', d['roofline']['family'], d['roofline']['avg_launch_us'], d['roofline']['step']['frac'])" >> gpurun_out/r06i_ab.txt 2>&1
}
for i in 1 2 3; do ab "NNR_BX3=0 NNR_POOL_TEAM=0" ""; ab "NNR_BX3=1" ""; done
for b in 8 16 32; do ab "NNR_BX3=0 NNR_POOL_TEAM=0" "--batch_size $b"; ab "NNR_BX3=1" "--batch_size $b"; done
ab "NNR_BX3=0 NNR_POOL_TEAM=0" "--config mhsa"; ab "NNR_BX3=1" "--config mhsa"
cat gpurun_out/r06i_ab.txt
