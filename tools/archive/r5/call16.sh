#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rm -rf /tmp/rb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rb -- python3 $ROOT/tools/r5/rocblas_names.py > /tmp/rb.log 2>&1
find /tmp/rb -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-400 | head -12 > $ROOT/gpurun_out/r05r_rocblas_names.txt
cat $ROOT/gpurun_out/r05r_rocblas_names.txt
