#!/bin/bash
# round 5, GPU call 11+: bf16x3 weight-gradient (TN) micro-benchmark, variants given in $1 (default: all), slices per shape chosen to fill 256 CUs for the 1-workgroup-per-CU forms
mkdir -p gpurun_out
V=${1:-"0 1 2 3 4 5 6 8 10 11"}
OUT=gpurun_out/r05m_bf16x3_tn.txt
rm -f $OUT
for shape in "400 400 112640 32" "400 400 112640 64" "200 400 112640 64" "832 200 112640 36" "1664 300 112640 20" "400 400 450560 64" "900 900 4352 8"; do
  for v in $V; do
    timeout 120 tools/micro/bf16x3_tn $shape $v 2>&1 | grep -E "variant" | grep -v JSON >> $OUT
  done
  timeout 120 tools/micro/bf16x3_tn $shape 0 2>&1 | grep native >> $OUT
done
cat $OUT
