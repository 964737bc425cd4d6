#!/bin/bash
# round 5, GPU call 15: one-wave NT tiles for the SUE launches (verdict item 1b): 256 x 64 / 384 x 64 against the 128 x 80 families, alone
mkdir -p gpurun_out
SHAPES=sue TILES=9,15,16,31,43,44,45,12 ROUNDS=7 timeout 300 python tools/gemm_pipe_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05q_sue_tiles.txt
