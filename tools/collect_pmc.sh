#!/bin/bash
# PMC traffic passes only (see collect_profiles.sh): tools/collect_pmc.sh r03a  ->  gpurun_out/profiles_r03a/pmc_traffic.{json,txt}
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
CMD="python3 $ROOT/bench.py --steps 6 --warmup 3 --no_cpu_baseline --no_isolated --sustained_seconds 0"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
export NNR_REPLAY=${NNR_REPLAY:-0}
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 700 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$TAG/$C -- $CMD > $OUT/bench_under_pmc_$C.json 2> $OUT/pmc_$C.err
  echo "$C rc=$?"
done
FF=$(find /tmp/pmc_$TAG/FETCH_SIZE -name "*counter_collection.csv" | head -1)
FW=$(find /tmp/pmc_$TAG/WRITE_SIZE -name "*counter_collection.csv" | head -1)
cd $ROOT
python3 tools/pmc_traffic.py $FF $FW $OUT/pmc_traffic.json 9 > $OUT/pmc_traffic.txt 2>&1
tail -20 $OUT/pmc_traffic.txt
