#!/bin/bash
# round 6, GPU call 27: the MHSA+MHSA step (configs[1], 1.96 ms): host-ahead check, per-call timeline, solo kernel table
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
ROOT=$PWD
O=$ROOT/gpurun_out
NNR_BENCH_STEP_MARKS=1 timeout 300 python bench.py --config mhsa --no_cpu_baseline --no_secondary --no_isolated --steps 20 --sustained_seconds 0 2> $O/r06A_marks.txt > $O/r06A_bench.json
grep -E "step marks|host enqueue" $O/r06A_marks.txt
timeout 300 python tools/tape_timeline.py --news_encoder MHSA --user_encoder MHSA > $O/r06A_timeline_mhsa.txt 2>&1
head -3 $O/r06A_timeline_mhsa.txt | cut -c1-300; tail -2 $O/r06A_timeline_mhsa.txt
MH="python3 $ROOT/bench.py --config mhsa --steps 12 --warmup 4 --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 0"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_m
NNR_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m/one -- $MH > $O/r06A_bench_one.json 2> $O/r06A_one.err
F=$(find /tmp/prof_m/one -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/r06A_mhsa_one_stream_kernel_stats.csv
head -25 $O/r06A_mhsa_one_stream_kernel_stats.csv | cut -c1-160
