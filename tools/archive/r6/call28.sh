#!/bin/bash
# round 6, GPU call 28: mid-size NT GEMMs of the MHSA user encoder alone, per tile
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
timeout 600 python tools/gemm_mid_shapes.py 3200 > gpurun_out/r06B_mid.txt 2>&1
timeout 600 python tools/gemm_mid_shapes.py 3520 >> gpurun_out/r06B_mid.txt 2>&1
cat gpurun_out/r06B_mid.txt
