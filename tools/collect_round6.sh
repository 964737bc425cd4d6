#!/bin/bash
# Round 6: the judged artefacts of the FINAL build in one gpurun call (from the repo root):  bash tools/collect_round6.sh r06i
#  1. default bench line (headline + secondary legs + CPU baseline); rocprofv3 --kernel-trace --stats of the headline command, in-step and with
#     every stream collapsed into one (NNR_ONE_STREAM=1) -> profiles/kernel_stats.json (build-id stamped; bench.py `roofline.rocprof` quotes it);
#     PMC FETCH_SIZE / WRITE_SIZE passes (NNR_REPLAY=0: the call-by-call native step -- same kernels, same order) -> profiles/pmc_traffic.json
#  2. the default line again with those files in place (roofline.traffic + roofline.rocprof quoted for this build id)
#  3. --prebuilt, per-GPU batch sweep -> profiles/batch_sweep.json, --config mhsa (+ its kernel table and matrix-pipe busy counter pass),
#     kernel tables of the batch-8 shard and of the V = 130 000 shard (round-5 verdict item 4 v)
#  4. per-call timelines of a replayed step (batch 64 / 8), phase table, kernel trace folded per step, 1 500-step soak
TAG=${1:-r06i}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export PYTHONWARNINGS=ignore
O=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $O
python3 bench.py > $O/bench_first.json 2> $O/bench_first.err
CMD="python3 $ROOT/bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 0"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/kt -- $CMD > $O/bench_under_kernel_trace.json 2> $O/kt.err
F=$(find /tmp/prof_$TAG/kt -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/kernel_stats.csv
T=$(find /tmp/prof_$TAG/kt -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && python3 $ROOT/tools/trace_steps.py $T --steps 6 --min_us 10 > $O/trace_steps_b64.txt 2>&1
NNR_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/one -- $CMD > $O/bench_one_stream_traced.json 2> $O/one.err
F1=$(find /tmp/prof_$TAG/one -name "*kernel_stats.csv" | head -1); [ -n "$F1" ] && cp $F1 $O/one_stream_kernel_stats.csv
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
python3 tools/kernel_stats_json.py $O/kernel_stats.csv $O/one_stream_kernel_stats.csv $O/kernel_stats.json > $O/kernel_stats.txt 2>&1
cp $O/kernel_stats.json profiles/kernel_stats.json 2>/dev/null
# MHSA + MHSA (BASELINE configs[1]): kernel table, matrix-pipe busy counters, then the bench line that quotes them
MH="python3 $ROOT/bench.py --config mhsa --steps 12 --warmup 4 --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/mhsa -- $MH > $O/bench_mhsa_traced.json 2> $O/mhsa_kt.err
F=$(find /tmp/prof_$TAG/mhsa -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/mhsa_kernel_stats.csv
NNR_REPLAY=0 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_${TAG}_mfma -- python3 $ROOT/bench.py --config mhsa --steps 6 --warmup 3 --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 0 > $O/bench_mhsa_under_pmc.json 2> $O/mfma_pmc.err
# the per-GPU shards of configs[3] (batch 8) and configs[4] (batch 16, V = 130 000, MIND-large dropout 0.1): kernel tables
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/b8 -- python3 $ROOT/bench.py --batch_size 8 --steps 16 --warmup 6 --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 0 > $O/bench_b8_traced.json 2> $O/b8_kt.err
F=$(find /tmp/prof_$TAG/b8 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/b8_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/v130 -- python3 $ROOT/bench.py --batch_size 16 --vocabulary_size 130000 --steps 16 --warmup 6 --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 0 > $O/bench_b16_v130000_traced.json 2> $O/v130_kt.err
F=$(find /tmp/prof_$TAG/v130 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/b16_v130000_kernel_stats.csv
cd $ROOT
python3 tools/pmc_mfma_busy.py /tmp/prof_${TAG}_mfma $O/pmc_mfma_busy.json > $O/pmc_mfma_busy.txt 2>&1
cp $O/pmc_mfma_busy.json profiles/pmc_mfma_busy.json 2>/dev/null
B="python3 bench.py --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 2"
$B --prebuilt > $O/bench_prebuilt.json 2>> $O/bench2.err
for b in 8 16 32 128; do $B --batch_size $b --steps 40 --warmup 8 > $O/bench_b$b.json 2>> $O/bench2.err; done
$B --steps 40 --warmup 8 > $O/bench_b64.json 2>> $O/bench2.err
NNR_BX3=0 NNR_POOL_TEAM=0 $B --steps 40 --warmup 8 > $O/bench_b64_f32_old_pools.json 2>> $O/bench2.err
$B --config mhsa > $O/bench_mhsa.json 2>> $O/bench2.err
python3 - <<PY
import json, sys
sys.path.insert(0, '$ROOT')
from nnr_amd import _lib
ms = {}
for b in (8, 16, 32, 64, 128):
    try:
        d = json.loads([l for l in open('$O/bench_b%d.json' % b) if l.startswith('{')][-1])
        ms[str(b)] = d['ms_per_step']
    except Exception as e:
        print('batch', b, 'FAILED', e)
json.dump({'build_id': _lib.build_id(), 'ms_per_step': ms, 'how': 'bench.py --batch_size B --steps 40 --warmup 8 --no_secondary on one MI355X (un-instrumented 40-step window)'},
          open('$O/batch_sweep.json', 'w'), indent=1)
print(ms)
PY
cp $O/batch_sweep.json profiles/batch_sweep.json 2>/dev/null
python3 bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python3 tools/tape_timeline.py --batch_size 64 > $O/timeline_b64.txt 2>&1
timeout 300 python3 tools/tape_timeline.py --batch_size 8 > $O/timeline_b8.txt 2>&1
timeout 300 python3 tools/phase_table.py --json $O/phase_b64.json > $O/phase_b64.txt 2>&1
timeout 300 python3 tools/pool_bench.py --stream content > $O/pool_alone.txt 2>&1
timeout 300 python3 tools/pool_bench.py --stream title >> $O/pool_alone.txt 2>&1
timeout 300 python3 tools/dyn_pair_bench.py > $O/nt_gemms_alone.txt 2>&1
NNR_BX3=0 timeout 300 python3 tools/dyn_pair_bench.py >> $O/nt_gemms_alone.txt 2>&1
timeout 600 python3 tools/replay_soak.py --steps 1500 > $O/soak.json 2> $O/soak.err
tail -c 300 $O/soak.json
python3 - <<PY
import json
for n in ['bench_first', 'bench', 'bench_prebuilt', 'bench_b8', 'bench_b16', 'bench_b32', 'bench_b64', 'bench_b64_f32_old_pools', 'bench_b128', 'bench_mhsa']:
    try:
        d = json.loads([l for l in open('$O/%s.json' % n) if l.startswith('{')][-1])
        r = d['roofline']
        print(n, d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step'), 'dominant', r['family'], r['frac'], 'rocprof', (r.get('rocprof') or {}).get('frac_in_step'), (r.get('rocprof') or {}).get('frac_solo'), 'step', r.get('step'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
        for k, v in (d.get('secondary') or {}).items():
            print('   secondary', k, v.get('ms_per_step'), v.get('value'), (v.get('step') or {}).get('frac'), v.get('error'))
    except Exception as e:
        print(n, 'FAILED', e)
PY
ls $O | head -80
