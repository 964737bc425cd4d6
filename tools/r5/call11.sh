#!/bin/bash
# round 5, GPU call 11: bf16x3 weight-gradient (TN) micro-benchmark, tile / accumulator / pitch variants
mkdir -p gpurun_out
rm -f gpurun_out/r05m_*.txt
for shape in "400 400 112640 46" "200 400 112640 80" "832 200 112640 50" "1664 300 112640 79" "400 400 450560 46" "900 900 4352 17"; do
  for v in 0 1 2 3 4 5 6 8; do
    timeout 120 tools/micro/bf16x3_tn $shape $v 2>&1 | grep -E "variant|JSON" | grep -v JSON >> gpurun_out/r05m_bf16x3_tn.txt
  done
  timeout 120 tools/micro/bf16x3_tn $shape 0 2>&1 | grep native >> gpurun_out/r05m_bf16x3_tn.txt
done
cat gpurun_out/r05m_bf16x3_tn.txt
