#!/bin/bash
# round 6, GPU call 58: matrix-pipe busy counter of the HEADLINE step's kernels (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE; the committed pass is of the MHSA step)
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_mb
NNR_REPLAY=0 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_mb -- python3 $ROOT/bench.py --steps 6 --warmup 3 --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 0 > $ROOT/gpurun_out/r06Y_bench_under_pmc.json 2> $ROOT/gpurun_out/r06Y_pmc.err
cd $ROOT
python3 tools/pmc_mfma_busy.py /tmp/prof_mb gpurun_out/r06Y_pmc_mfma_busy_headline.json > gpurun_out/r06Y_pmc_mfma_busy_headline.txt 2>&1
head -30 gpurun_out/r06Y_pmc_mfma_busy_headline.txt | cut -c1-200
