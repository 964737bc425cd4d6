#!/bin/bash
# Round 4: the judged artefacts of the FINAL build (run through gpurun from the repo root):  tools/r4_collect.sh r04f
#  1. default bench line (with the CPU baseline leg) + rocprofv3 --kernel-trace --stats of the same command + PMC FETCH_SIZE / WRITE_SIZE
#     passes (tools/collect_profiles.sh) -> pmc_traffic.json with this build's id
#  2. bench again with that pmc_traffic.json in place (roofline.traffic / roofline.hbm[*].traffic quoted), --prebuilt, per-GPU batch sweep
#  3. --config mhsa + its matrix-pipe busy counter pass (tools/pmc_mfma_busy.py)
#  4. one-stream kernel stats (solo durations), per-call timelines of a replayed step (batch 64 / 8), 1 500-step soak
TAG=${1:-r04f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $O
bash tools/collect_profiles.sh $TAG > $O/collect.log 2>&1
cp $O/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
python3 bench.py > $O/bench_with_traffic.json 2> $O/bench_with_traffic.err
B="python3 bench.py --no_cpu_baseline --no_isolated --sustained_seconds 2"
$B --prebuilt > $O/bench_prebuilt.json 2>> $O/bench.err
for b in 8 16 32 128; do $B --batch_size $b --steps 40 --warmup 8 > $O/bench_b$b.json 2>> $O/bench.err; done
$B --dense > $O/bench_dense.json 2>> $O/bench.err
# MHSA + MHSA (BASELINE configs[1]) and the matrix-pipe busy counters of its kernels
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_${TAG}_mfma
NNR_REPLAY=0 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_${TAG}_mfma -- python3 $ROOT/bench.py --config mhsa --steps 6 --warmup 3 --no_cpu_baseline --no_isolated --sustained_seconds 0 > $O/bench_mhsa_under_pmc.json 2> $O/mfma_pmc.err
cd $ROOT
python3 tools/pmc_mfma_busy.py /tmp/prof_${TAG}_mfma $O/pmc_mfma_busy.json > $O/pmc_mfma_busy.txt 2>&1
cp $O/pmc_mfma_busy.json profiles/pmc_mfma_busy.json 2>/dev/null
$B --config mhsa > $O/bench_mhsa.json 2>> $O/bench.err
# solo kernel durations
cd /tmp
rm -rf /tmp/prof_${TAG}_one
NNR_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_one -- python3 $ROOT/bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_isolated --sustained_seconds 0 > $O/bench_one_stream_traced.json 2> $O/one.err
F=$(find /tmp/prof_${TAG}_one -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/one_stream_kernel_stats.csv
cd $ROOT
timeout 300 python3 tools/tape_timeline.py --batch_size 64 > $O/timeline_b64.txt 2>&1
timeout 300 python3 tools/tape_timeline.py --batch_size 8 > $O/timeline_b8.txt 2>&1
timeout 600 python3 tools/replay_soak.py --steps 1500 > $O/soak.json 2> $O/soak.err
tail -3 $O/soak.json
python3 - <<PY
import json
for n in ['bench', 'bench_with_traffic', 'bench_prebuilt', 'bench_b8', 'bench_b16', 'bench_b32', 'bench_b128', 'bench_dense', 'bench_mhsa']:
    try:
        d = json.loads([l for l in open('$O/%s.json' % n) if l.startswith('{')][-1])
        r = d['roofline']
        print(n, d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step'), 'dominant', r['family'], r['frac'], 'traffic', r.get('traffic'), 'step', r.get('step'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e:
        print(n, 'FAILED', e)
PY
ls -la $O | head -50
