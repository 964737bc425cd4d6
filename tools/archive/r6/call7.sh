#!/bin/bash
# round 6, GPU call 7: bf16x3 NT kernel with the truncation split + lean DMA issue (unit tests, alone at the step's shapes, in the step), and the
# packed pool variants inside the step (NNR_POOL_TEAM bits, NNR_POOL_R), un-instrumented, three interleaved rounds
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
(timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q --tb=short -x -k "gemm or pool" 2>&1 | grep -v amdgpu.ids | tail -12) > gpurun_out/r06g_tests.log
tail -4 gpurun_out/r06g_tests.log
NNR_BX3=1 timeout 300 python tools/dyn_pair_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06g_dyn_pair.txt
rm -f gpurun_out/r06g_ab.txt
ab() {
  echo -n "$1 : " >> gpurun_out/r06g_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['roofline']['hbm']; print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], 'pool_bwd', h['pool_bwd']['avg_launch_us'], 'pool_fwd', h['pool_fwd']['avg_launch_us'], 'dominant', d['roofline']['family'], d['roofline']['avg_launch_us'])" >> gpurun_out/r06g_ab.txt 2>&1
}
for i in 1 2 3; do
  ab "NNR_POOL_TEAM=0"
  ab "NNR_POOL_TEAM=3 NNR_POOL_R=16"
  ab "NNR_POOL_TEAM=2 NNR_POOL_R=16"
  ab "NNR_POOL_TEAM=3 NNR_POOL_R=8"
  ab "NNR_POOL_TEAM=2 NNR_POOL_R=8"
done
ab "NNR_POOL_TEAM=0 NNR_BX3=0"
cat gpurun_out/r06g_ab.txt
