#!/bin/bash
# Round 4, GPU call 3: the whole affected GPU suite (no -x), scatter micro-benchmark, bench A/B, one-stream kernel stats.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
O=gpurun_out/r04c
mkdir -p $O
( timeout 1800 python -m pytest tests/test_hip_ops_gpu.py tests/test_hip_headline_gpu.py tests/test_hip_tape_gpu.py tests/test_hip_dropout_gpu.py tests/test_hip_model_gpu.py tests/test_hip_layers_gpu.py tests/test_hip_dp_gpu.py tests/test_hip_edge_gpu.py -q --durations=40 ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/tests.log | tail -30
timeout 300 python tools/scatter_bench.py > $O/scatter_bench.txt 2>&1; cat $O/scatter_bench.txt | tail -4
B="python bench.py --no_cpu_baseline --no_isolated --sustained_seconds 2 --prebuilt"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"; }
run default NNR_X=0
run nosort NNR_SCATTER_SORTED=0
run nodet NNR_TN_SLAB=0 NNR_SCATTER_SORTED=0 NNR_DETERMINISTIC=0
run default2 NNR_X=0
timeout 300 $B --batch_size 8 > $O/bench_b8.json 2> $O/bench_b8.err; echo "b8 rc=$?"
NNR_TN_SLAB=0 NNR_SCATTER_SORTED=0 NNR_DETERMINISTIC=0 timeout 300 $B --batch_size 8 > $O/bench_b8_nodet.json 2> $O/bench_b8_nodet.err
timeout 300 python bench.py --no_cpu_baseline --no_isolated --sustained_seconds 2 > $O/bench_devcorpus.json 2> $O/bench_devcorpus.err
# solo kernel durations: every stream collapsed into one, under the kernel tracer
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r04c
NNR_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r04c/one -- python3 $R/bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_isolated --sustained_seconds 0 --prebuilt > $R/$O/bench_one_stream_traced.json 2> $R/$O/one.err
F=$(find /tmp/prof_r04c/one -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/$O/one_stream_kernel_stats.csv
cd $R
python - <<'PY'
import json, glob, csv
for f in sorted(glob.glob('gpurun_out/r04c/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        r = d.get('roofline') or {}
        print('%-28s %8.1f imp/s %7.3f ms  sustained %s  step %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step'), (r.get('step') or {})))
        fam = r.get('families') or {}
        print('    ' + '  '.join('%s %.0fus x%d' % (k.replace('gemm_', ''), 1000 * v['ms'] / max(1, v['launches']), v['launches']) for k, v in list(fam.items())[:9]))
        if r.get('hbm'):
            print('    hbm: ' + '  '.join('%s %.0fus %.0fGB/s' % (k, v['avg_launch_us'], v['achieved']) for k, v in r['hbm'].items()))
    except Exception as e:
        print(f, 'unreadable', e)
try:
    rows = list(csv.DictReader(open('gpurun_out/r04c/one_stream_kernel_stats.csv')))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('one-stream kernel stats: total %.2f ms over 16 steps = %.3f ms/step' % (tot / 1e6, tot / 16e6))
    for r in rows[:32]:
        print('  %-70s %5d calls %9.1f us avg %5.1f %%' % (r['Name'][:70], int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage'])))
except Exception as e:
    print('no kernel stats', e)
PY
