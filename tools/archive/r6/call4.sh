#!/bin/bash
# round 6, GPU call 4: register-resident packed pools (NNR_POOL_TEAM): unit tests, alone (both streams, both forms, results compared), in the step
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
(timeout 900 python -m pytest tests/test_hip_ops_gpu.py tests/test_hip_layers_gpu.py -m gpu -q --tb=short -x -k "pool or bf16x3 or edge or mhsa or attention" 2>&1 | grep -v amdgpu.ids | tail -12) > gpurun_out/r06d_tests.log
tail -5 gpurun_out/r06d_tests.log
rm -f gpurun_out/r06d_pool.txt
for st in content title; do
  for t in 0 1; do
    NNR_POOL_TEAM=$t timeout 200 python tools/pool_bench.py --stream $st --dump /tmp/pool_${st}_$t.pt 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06d_pool.txt
  done
  python - >> gpurun_out/r06d_pool.txt 2>&1 <<PY
import torch
a, b = torch.load('/tmp/pool_${st}_0.pt'), torch.load('/tmp/pool_${st}_1.pt')
print('  ${st}: max |team - stream| relative to max |stream|:', {k: float((a[k] - b[k]).abs().max() / a[k].abs().max().clamp_min(1e-30)) for k in a})
PY
done
cat gpurun_out/r06d_pool.txt
rm -f gpurun_out/r06d_ab.txt
ab() {
  echo -n "$1 $2: " >> gpurun_out/r06d_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['roofline']['hbm']; print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], 'pool_bwd', h['pool_bwd']['avg_launch_us'], h['pool_bwd']['frac'], 'pool_fwd', h['pool_fwd']['avg_launch_us'], h['pool_fwd']['frac'])" >> gpurun_out/r06d_ab.txt 2>&1
}
for i in 1 2 3; do ab "NNR_POOL_TEAM=0" ""; ab "NNR_POOL_TEAM=1" ""; done
ab "NNR_POOL_TEAM=0" "--batch_size 8"; ab "NNR_POOL_TEAM=1" "--batch_size 8"
ab "NNR_POOL_TEAM=0" "--config mhsa"; ab "NNR_POOL_TEAM=1" "--config mhsa"
cat gpurun_out/r06d_ab.txt
