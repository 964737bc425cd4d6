#!/bin/bash
# Extra evidence lines on the final build (no code change): device-corpus variant, dense worst case, and solo kernel durations
# (every HIP stream of the step collapsed into one: NNR_ONE_STREAM=1) under rocprofv3 --kernel-trace --stats.   tools/r3_extra_evidence.sh TAG
TAG=${1:-r03g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd $ROOT
timeout 200 python3 bench.py --device_corpus --no_cpu_baseline --no_isolated --sustained_seconds 1 > gpurun_out/${TAG}_bench_device_corpus.json 2> gpurun_out/${TAG}.err
timeout 200 python3 bench.py --dense --no_cpu_baseline --no_isolated --sustained_seconds 1 > gpurun_out/${TAG}_bench_dense.json 2>> gpurun_out/${TAG}.err
cd /tmp && export TMPDIR=/tmp
export NNR_ONE_STREAM=1
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -- python3 $ROOT/bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_isolated --sustained_seconds 0 > $ROOT/gpurun_out/${TAG}_bench_one_stream.json 2>> $ROOT/gpurun_out/${TAG}.err
F=$(find /tmp/kt1 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $ROOT/gpurun_out/${TAG}_one_stream_kernel_stats.csv
unset NNR_ONE_STREAM
cd $ROOT
python3 - <<PY
import json
for n in ['device_corpus', 'dense', 'one_stream']:
    try:
        d = json.load(open('gpurun_out/${TAG}_bench_%s.json' % n))
        print(n, d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step'), d['roofline']['step'])
    except Exception as e:
        print(n, 'FAILED', e)
PY
tail -3 gpurun_out/${TAG}.err
