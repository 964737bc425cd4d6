#!/bin/bash
# round 6, GPU call 6: dead-workgroup cost of capacity-sized launches under co-residency (tools/dyn_pair_bench.py), bf16x3 and fp32 kernels
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
for r in 1 0; do NNR_BX3=$r timeout 300 python tools/dyn_pair_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06f_dyn_pair.txt
