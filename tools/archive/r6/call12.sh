#!/bin/bash
# round 6, GPU call 12: full GPU suite on the pruned tree; three-stage bf16x3 tiles alone at the step's shapes
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
(timeout 2000 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v amdgpu.ids | tail -15) > gpurun_out/r06l_tests.log
tail -5 gpurun_out/r06l_tests.log | cut -c1-300
for t in 50 53 54 55; do echo "NNR_BX3_TILE=$t"; NNR_BX3_TILE=$t timeout 300 python tools/dyn_pair_bench.py 2>&1 | grep -v amdgpu.ids | grep "capacity\|^N"; done | tee gpurun_out/r06l_tiles.txt
