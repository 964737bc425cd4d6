#!/bin/bash
# round 6, GPU call 26: is the head of the step paced by the HOST's enqueue order?  host-ahead diagnostic + enqueue-order A/B (content chain first, token sorts behind both projections)
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
ROOT=$PWD
O=$ROOT/gpurun_out
rm -f $O/r06z_ab.txt
NNR_BENCH_STEP_MARKS=1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 20 --sustained_seconds 0 2> $O/r06z_marks.txt > /dev/null
grep -E "step marks|host enqueue" $O/r06z_marks.txt
ab() {
  echo -n "$1 : " >> $O/r06z_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> $O/r06z_ab.txt 2>&1
}
for i in 1 2 3; do ab "NNR_X=0"; ab "NNR_CONTENT_FIRST=1"; ab "NNR_TSORT_LATE=1"; ab "NNR_CONTENT_FIRST=1 NNR_TSORT_LATE=1"; done
cat $O/r06z_ab.txt
CMD="python3 $ROOT/bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 0"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_h
export NNR_CONTENT_FIRST=1 NNR_TSORT_LATE=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_h/kt -- $CMD > $O/r06z_bench_traced.json 2> $O/r06z_kt.err
T=$(find /tmp/prof_h/kt -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_steps.py $T --steps 6 --min_us 0 > $O/r06z_trace_all.txt 2>&1
head -100 $O/r06z_trace_all.txt | grep -v "split_bf16x3\|rocprim"
