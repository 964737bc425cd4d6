#!/bin/bash
# round 5, GPU call 13: ablations of the TN bf16x3 v3 loop (which component sets the stage time)
mkdir -p gpurun_out
OUT=gpurun_out/r05o_bf16x3_tn_ablation.txt
rm -f $OUT
for shape in "400 400 112640 32" "832 200 112640 36"; do
  for v in ${1:-10 20 21 22 23}; do
    timeout 120 tools/micro/bf16x3_tn $shape $v 2>&1 | grep -E "variant|split pass" | grep -v JSON | sed -E "s/\(([^;)]{0,60})[^)]*\)/(\1)/" >> $OUT
  done
done
cat $OUT
