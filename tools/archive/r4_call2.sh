#!/bin/bash
# Round 4, GPU call 2: reproducible gradients (split-K slabs, sorted segmented embedding gradient, fixed-order column sums) -- unit
# tests, the headline-size parity / determinism tests, and the bench A/B of each piece.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04b
mkdir -p $O
( timeout 1500 python -m pytest tests/test_hip_ops_gpu.py tests/test_hip_headline_gpu.py tests/test_hip_tape_gpu.py tests/test_hip_dropout_gpu.py tests/test_hip_model_gpu.py tests/test_hip_layers_gpu.py -q -x --durations=15 ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -30 $O/tests.log
B="python bench.py --no_cpu_baseline --no_isolated --sustained_seconds 2 --prebuilt"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"; }
run default NNR_X=0
run noslab NNR_TN_SLAB=0
run nosort NNR_SCATTER_SORTED=0
run nodet NNR_TN_SLAB=0 NNR_SCATTER_SORTED=0 NNR_DETERMINISTIC=0
run stages192 NNR_TN_STAGES=192
run stages384 NNR_TN_STAGES=384
run want256 NNR_TN_WANT=256 NNR_TN_STAGES=384
run default2 NNR_X=0
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04b/bench_*.json')):
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
PY
