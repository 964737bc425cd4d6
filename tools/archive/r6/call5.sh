#!/bin/bash
# round 6, GPU call 5: why is the step slower with the faster pool kernels?  rocprofv3 kernel traces (hardware timestamps, no per-call events) of the
# replayed step with NNR_POOL_TEAM=0 / 1, folded per step by tools/trace_steps.py; + the edge-value test
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
(timeout 300 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q --tb=short -x -k "edge" 2>&1 | grep -v amdgpu.ids | tail -12) > gpurun_out/r06e_tests.log
tail -3 gpurun_out/r06e_tests.log
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for t in 0 1; do
  rm -rf /tmp/tr_$t
  NNR_POOL_TEAM=$t timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$t -- python3 $ROOT/bench.py --steps 16 --warmup 6 --no_cpu_baseline --no_isolated --no_secondary --sustained_seconds 0 --roofline_steps 1 > $ROOT/gpurun_out/r06e_bench_$t.json 2> $ROOT/gpurun_out/r06e_err_$t.txt
  F=$(find /tmp/tr_$t -name "*kernel_trace.csv" | head -1)
  python3 $ROOT/tools/trace_steps.py $F --steps 8 --min_us 10 > $ROOT/gpurun_out/r06e_trace_team$t.txt 2>&1
  head -3 $F > $ROOT/gpurun_out/r06e_trace_head_$t.txt
done
cd $ROOT
head -2 gpurun_out/r06e_trace_team0.txt; head -2 gpurun_out/r06e_trace_team1.txt
python3 - <<'PY'
import json
for t in (0, 1):
    try:
        d = json.loads([l for l in open('gpurun_out/r06e_bench_%d.json' % t) if l.startswith('{')][-1])
        print('team', t, d['ms_per_step'], d['value'])
    except Exception as e:
        print('team', t, 'FAILED', e)
PY
