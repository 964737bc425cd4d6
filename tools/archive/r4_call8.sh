#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04h
mkdir -p $O
TN_TILES=26,27,30,32,37,38,39 ROUNDS=3 timeout 900 python tools/gemm_pipe_bench.py tn > $O/gemm_tn_bench.txt 2>&1; grep -v amdgpu.ids $O/gemm_tn_bench.txt
