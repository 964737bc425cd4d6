#!/bin/bash
# round 6, GPU call 53: per-call timeline of the batch-8 step with the exchange in it (dense table bucket / touched rows), C-ABI binding
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
NNR_DP_NATIVE=1 timeout 300 python tools/dp_step_timing.py --batch_size 8 --exchange 1 --timeline > gpurun_out/r06T_dp_dense.txt 2>&1
NNR_DP_NATIVE=1 NNR_DP_TOUCHED_ROWS=1 timeout 300 python tools/dp_step_timing.py --batch_size 8 --exchange 1 --timeline > gpurun_out/r06T_dp_touched.txt 2>&1
timeout 300 python tools/dp_step_timing.py --batch_size 8 --exchange 0 --timeline > gpurun_out/r06T_dp_none.txt 2>&1
grep "^batch" gpurun_out/r06T_dp_*.txt | cut -c1-120
