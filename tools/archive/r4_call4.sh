#!/bin/bash
# Round 4, GPU call 4: the WHOLE GPU suite (wall time matters: the driver's cap), scatter micro-benchmark after the load-hoisting fix,
# A/B of the gate-backward epilogue fusion and of the balanced backward recurrence, batch 8, MHSA native step.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04d
mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --durations=25 ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/tests.log | tail -30
timeout 300 python tools/scatter_bench.py > $O/scatter_bench.txt 2>&1; tail -3 $O/scatter_bench.txt
B="python bench.py --no_cpu_baseline --no_isolated --sustained_seconds 2 --prebuilt"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"; }
run default NNR_X=0
run nogatefuse NNR_GATE_FUSED=0
run nobal NNR_LSTM_BWD_BAL=0
run neither NNR_GATE_FUSED=0 NNR_LSTM_BWD_BAL=0
run default2 NNR_X=0
run nodet NNR_TN_SLAB=0 NNR_SCATTER_SORTED=0 NNR_DETERMINISTIC=0
timeout 300 $B --batch_size 8 > $O/bench_b8.json 2> $O/bench_b8.err; echo "b8 rc=$?"
NNR_TN_SLAB=0 NNR_SCATTER_SORTED=0 NNR_DETERMINISTIC=0 timeout 300 $B --batch_size 8 > $O/bench_b8_nodet.json 2> $O/bench_b8_nodet.err
timeout 300 $B --config mhsa > $O/bench_mhsa.json 2> $O/bench_mhsa.err; echo "mhsa rc=$?"
NNR_MHSA_NATIVE=0 timeout 300 $B --config mhsa > $O/bench_mhsa_autograd.json 2> $O/bench_mhsa_autograd.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04d/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        r = d.get('roofline') or {}
        print('%-28s %8.1f imp/s %7.3f ms  sustained %s  calls %s step %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step'), d['config'].get('abi_calls_per_step'), (r.get('step') or {})))
        fam = r.get('families') or {}
        print('    ' + '  '.join('%s %.0fus x%d' % (k.replace('gemm_', ''), 1000 * v['ms'] / max(1, v['launches']), v['launches']) for k, v in list(fam.items())[:9]))
        if r.get('hbm'):
            print('    hbm: ' + '  '.join('%s %.0fus %.0fGB/s' % (k, v['avg_launch_us'], v['achieved']) for k, v in r['hbm'].items()))
    except Exception as e:
        print(f, 'unreadable', e)
PY
