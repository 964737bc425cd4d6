#!/bin/bash
# Collect the round's judged artefacts on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r02a   ->  gpurun_out/profiles_r02a/{bench.json, kernel_stats.csv, pmc_traffic.json, ...}
# 1. the default bench line (with the CPU baseline leg), 2. rocprofv3 --kernel-trace --stats of the same command (CSV),
# 3. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the same command, aggregated by tools/pmc_traffic.py.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
CMD="python3 $ROOT/bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_isolated --sustained_seconds 0"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/kt -- $CMD > $OUT/bench_under_kernel_trace.json 2> $OUT/kt.err
F=$(find /tmp/prof_$TAG/kt -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/kernel_stats.csv
[ -n "$SKIP_PMC" ] && { ls -la $OUT; exit 0; }     # (tools/collect_pmc.sh collects the counter passes on their own)
# PMC passes: rocprofv3 serialises every dispatch; the call-by-call native step (NNR_REPLAY=0: same kernels, same order, issued from
# Python) is used here -- the natively replayed step, enqueued as a whole across four streams, did not finish under dispatch
# serialisation within 24 minutes (round 3)
export NNR_REPLAY=0
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$TAG/$C -- $CMD > $OUT/bench_under_pmc_$C.json 2> $OUT/pmc_$C.err
done
unset NNR_REPLAY
FF=$(find /tmp/prof_$TAG/FETCH_SIZE -name "*counter_collection.csv" | head -1)
FW=$(find /tmp/prof_$TAG/WRITE_SIZE -name "*counter_collection.csv" | head -1)
cd $ROOT
python3 tools/pmc_traffic.py $FF $FW $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
ls -la $OUT
