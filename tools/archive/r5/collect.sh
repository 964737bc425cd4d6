#!/bin/bash
# Round 5: the judged artefacts of the FINAL build in one gpurun call (from the repo root):  bash tools/r5/collect.sh r05f
#  1. default bench line (headline + secondary legs + CPU baseline) ; rocprofv3 --kernel-trace --stats of the headline command ;
#     PMC FETCH_SIZE / WRITE_SIZE passes (NNR_REPLAY=0: the call-by-call native step -- same kernels, same order) -> pmc_traffic.json
#  2. the default line again with that pmc_traffic.json in place (roofline.traffic quoted for this build id)
#  3. --prebuilt, per-GPU batch sweep, --config mhsa (+ its matrix-pipe busy counter pass)
#  4. one-stream kernel table (solo durations), per-call timelines of a replayed step (batch 64 / 16 / 8), 1 500-step soak
TAG=${1:-r05f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
CMD="python3 $ROOT/bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 0"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/kt -- $CMD > $O/bench_under_kernel_trace.json 2> $O/kt.err
F=$(find /tmp/prof_$TAG/kt -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/kernel_stats.csv
export NNR_REPLAY=0
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$TAG/$C -- $CMD > $O/bench_under_pmc_$C.json 2> $O/pmc_$C.err
done
unset NNR_REPLAY
FF=$(find /tmp/prof_$TAG/FETCH_SIZE -name "*counter_collection.csv" | head -1)
FW=$(find /tmp/prof_$TAG/WRITE_SIZE -name "*counter_collection.csv" | head -1)
cd $ROOT
python3 tools/pmc_traffic.py $FF $FW $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
cp $O/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
# MHSA + MHSA (BASELINE configs[1]): matrix-pipe busy counters of its kernels, then the bench line that quotes them
cd /tmp
rm -rf /tmp/prof_${TAG}_mfma
NNR_REPLAY=0 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_${TAG}_mfma -- python3 $ROOT/bench.py --config mhsa --steps 6 --warmup 3 --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 0 > $O/bench_mhsa_under_pmc.json 2> $O/mfma_pmc.err
cd $ROOT
python3 tools/pmc_mfma_busy.py /tmp/prof_${TAG}_mfma $O/pmc_mfma_busy.json > $O/pmc_mfma_busy.txt 2>&1
cp $O/pmc_mfma_busy.json profiles/pmc_mfma_busy.json 2>/dev/null
python3 bench.py > $O/bench_with_traffic.json 2> $O/bench_with_traffic.err
B="python3 bench.py --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 2"
$B --prebuilt > $O/bench_prebuilt.json 2>> $O/bench2.err
for b in 8 16 32 128; do $B --batch_size $b --steps 40 --warmup 8 > $O/bench_b$b.json 2>> $O/bench2.err; done
$B --config mhsa > $O/bench_mhsa.json 2>> $O/bench2.err
NNR_MHSA_PACKED=0 $B --config mhsa > $O/bench_mhsa_dense_rows.json 2>> $O/bench2.err
# solo kernel durations
cd /tmp
rm -rf /tmp/prof_${TAG}_one
NNR_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_one -- python3 $ROOT/bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 0 > $O/bench_one_stream_traced.json 2> $O/one.err
F=$(find /tmp/prof_${TAG}_one -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/one_stream_kernel_stats.csv
cd $ROOT
timeout 300 python3 tools/tape_timeline.py --batch_size 64 > $O/timeline_b64.txt 2>&1
timeout 300 python3 tools/tape_timeline.py --batch_size 16 > $O/timeline_b16.txt 2>&1
timeout 300 python3 tools/tape_timeline.py --batch_size 8 > $O/timeline_b8.txt 2>&1
timeout 600 python3 tools/replay_soak.py --steps 1500 > $O/soak.json 2> $O/soak.err
tail -c 400 $O/soak.json
python3 - <<PY
import json
for n in ['bench', 'bench_with_traffic', 'bench_prebuilt', 'bench_b8', 'bench_b16', 'bench_b32', 'bench_b128', 'bench_mhsa', 'bench_mhsa_dense_rows']:
    try:
        d = json.loads([l for l in open('$O/%s.json' % n) if l.startswith('{')][-1])
        r = d['roofline']
        print(n, d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step'), 'dominant', r['family'], r['frac'], 'traffic', r.get('traffic'), 'step', r.get('step'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
        for k, v in (d.get('secondary') or {}).items():
            print('   secondary', k, v.get('ms_per_step'), v.get('value'), v.get('step'), v.get('error'))
    except Exception as e:
        print(n, 'FAILED', e)
PY
ls $O | head -60
