#!/bin/bash
# round 6, GPU call 20: verdict item 1(b) gate -- the weight-gradient (TN) bf16x3 micro-benchmark with the truncation split (variants 40 / 41) against
# round 5's round-to-nearest split (10 / 42) and the library's fp32 TN kernels, at the step's live shapes (K = live tokens)
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/nnr_amd:$LD_LIBRARY_PATH
B=tools/micro/bf16x3_tn
rm -f gpurun_out/r06t_tn.txt
run() { timeout 120 $B $@ 2>&1 | grep -v "^JSON\|amdgpu.ids" | tail -2 >> gpurun_out/r06t_tn.txt; }
for Z in 46; do for v in 10 40; do echo "== 400 x 400 x 80000, $Z slices, variant $v" >> gpurun_out/r06t_tn.txt; run 400 400 80000 $Z $v; done; done
for Z in 64; do for v in 10 40; do echo "== 200 x 400 x 80000, $Z slices, variant $v" >> gpurun_out/r06t_tn.txt; run 200 400 80000 $Z $v; done; done
for Z in 50; do for v in 10 40; do echo "== 832 x 200 x 80000, $Z slices, variant $v" >> gpurun_out/r06t_tn.txt; run 832 200 80000 $Z $v; done; done
for Z in 20 40; do for v in 42 41 40; do echo "== 1664 x 300 x 80000, $Z slices, variant $v" >> gpurun_out/r06t_tn.txt; run 1664 300 80000 $Z $v; done; done
for Z in 12; do for v in 10 40; do echo "== 900 x 900 x 4352, $Z slices, variant $v" >> gpurun_out/r06t_tn.txt; run 900 900 4352 $Z $v; done; done
cat gpurun_out/r06t_tn.txt | cut -c1-260
