#!/bin/bash
# Round 4, GPU call 1: the changed / new GPU tests, the bench line with its new defaults, continuity with round 3 (--prebuilt),
# the 64 x 208 weight-gradient tile A/B, the MHSA+MHSA leg.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04a
mkdir -p $O
( timeout 900 python -m pytest tests/test_hip_tape_gpu.py "tests/test_hip_ops_gpu.py::test_gemm_pipelined_tn_tiles" -x -q -k "not tile20 and not tile21 and not tile22 and not tile23 and not tile24 and not tile25 and not tile28 and not tile29" ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
B="python bench.py --no_cpu_baseline --no_isolated --sustained_seconds 2"
timeout 300 $B > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
timeout 300 $B --prebuilt > $O/bench_prebuilt.json 2> $O/bench_prebuilt.err; echo "prebuilt rc=$?"
NNR_TN_T64=1 timeout 300 $B --prebuilt > $O/bench_prebuilt_t64_1.json 2> $O/bench_prebuilt_t64_1.err; echo "t64=1 rc=$?"
NNR_TN_T64=3 timeout 300 $B --prebuilt > $O/bench_prebuilt_t64_3.json 2> $O/bench_prebuilt_t64_3.err; echo "t64=3 rc=$?"
timeout 300 $B --prebuilt > $O/bench_prebuilt_again.json 2> $O/bench_prebuilt_again.err; echo "prebuilt again rc=$?"
timeout 300 $B --config mhsa > $O/bench_mhsa.json 2> $O/bench_mhsa.err; echo "mhsa rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04a/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        r = d.get('roofline') or {}
        print('%-40s %8.1f imp/s %7.3f ms  sustained %s  step %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step'), (r.get('step') or {})))
        fam = r.get('families') or {}
        print('    ' + '  '.join('%s %.0fus x%d' % (k.replace('gemm_', ''), 1000 * v['ms'] / max(1, v['launches']), v['launches']) for k, v in list(fam.items())[:9]))
        if r.get('hbm'):
            print('    hbm: ' + '  '.join('%s %.0fus %.0fGB/s' % (k, v['avg_launch_us'], v['achieved']) for k, v in r['hbm'].items()))
        if r.get('mhsa'):
            print('    mhsa:', r['mhsa'])
    except Exception as e:
        print(f, 'unreadable', e)
PY
