#!/bin/bash
# round 6, GPU call 25: un-instrumented hardware-timestamp trace of the step's head, every kernel (why does the content projection start ~400 us after its gather?)
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
ROOT=$PWD
O=$ROOT/gpurun_out
CMD="python3 $ROOT/bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 0"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_h
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_h/kt -- $CMD > $O/r06y_bench_traced.json 2> $O/r06y_kt.err
T=$(find /tmp/prof_h/kt -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_steps.py $T --steps 6 --min_us 0 > $O/r06y_trace_all.txt 2>&1
head -130 $O/r06y_trace_all.txt
