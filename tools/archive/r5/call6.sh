#!/bin/bash
# round 5, GPU call 6: bf16x3 micro-benchmark, version 2 (LDS-DMA staging, A split in registers)
mkdir -p gpurun_out
rm -f gpurun_out/r05h_bf16x3.txt
for shape in "450560 400 400" "450560 200 400" "450560 400 200" "131072 1664 304" "131072 304 1664" "4352 900 904"; do
  timeout 300 tools/micro/bf16x3_gemm $shape 2>&1 | tee -a gpurun_out/r05h_bf16x3.txt
done
