#!/bin/bash
# Round 4, GPU call 5: wide NT tiles (isolated + in-step A/B), the whole GPU suite again (wall time with 16 oracle threads).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04e
mkdir -p $O
TILES=15,9,33,34,36,18 ROUNDS=3 timeout 600 python tools/gemm_pipe_bench.py > $O/gemm_pipe_bench.txt 2>&1; cat $O/gemm_pipe_bench.txt | grep -v amdgpu.ids
B="python bench.py --no_cpu_baseline --no_isolated --sustained_seconds 2 --prebuilt"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"; }
run default NNR_X=0
run proj33 NNR_PROJ_TILE=33
run proj34 NNR_PROJ_TILE=34
run proj36 NNR_PROJ_TILE=36
run gate33 NNR_GATE_TILE=33
run both33 NNR_PROJ_TILE=33 NNR_GATE_TILE=33
run default2 NNR_X=0
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04e/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        r = d.get('roofline') or {}
        print('%-28s %8.1f imp/s %7.3f ms  sustained %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step')))
        fam = r.get('families') or {}
        print('    ' + '  '.join('%s %.0fus x%d' % (k.replace('gemm_', ''), 1000 * v['ms'] / max(1, v['launches']), v['launches']) for k, v in list(fam.items())[:10]))
    except Exception as e:
        print(f, 'unreadable', e)
PY
( timeout 1500 python -m pytest tests -m gpu -q --durations=12 ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|FAILED|ERROR|rc=|ReLU-kink" $O/tests.log | tail -12
grep -n "slowest" -A13 $O/tests.log
