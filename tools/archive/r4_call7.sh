#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04g
mkdir -p $O
( timeout 900 python -m pytest tests/test_hip_headline_gpu.py tests/test_hip_tape_gpu.py tests/test_hip_dropout_gpu.py -q -x ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/tests.log | tail -8
B="python bench.py --no_cpu_baseline --no_isolated --sustained_seconds 2 --prebuilt"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"; }
run default NNR_X=0
run suejoin NNR_SUE_JOIN=1
run default2 NNR_X=0
run suejoin2 NNR_SUE_JOIN=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04g/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print('%-28s %8.1f imp/s %7.3f ms  sustained %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step')))
    except Exception as e:
        print(f, 'unreadable', e)
PY
