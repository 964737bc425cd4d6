#!/bin/bash
# round 6, GPU call 3: full GPU suite on the bf16x3-default tree (+ new DP replay / edge-value tests), then the driver's bench command
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
(timeout 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v amdgpu.ids | tail -25) > gpurun_out/r06c_tests.log
tail -8 gpurun_out/r06c_tests.log
timeout 600 python bench.py > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06c_bench.json') if l.startswith('{')][-1])
r = d['roofline']
print(d['value'], d['ms_per_step'], d['sustained'], 'dominant', r['family'], r['frac'], r['avg_launch_us'], 'step', r['step'])
print('rocprof', r.get('rocprof'))
print('matrix_path', {k: v for k, v in d['config']['matrix_path'].items() if k != 'launch_classes'})
for k, v in (d.get('secondary') or {}).items():
    print('   secondary', k, v.get('ms_per_step'), v.get('value'), v.get('step'), v.get('error'), v.get('leg_seconds'))
print('hbm', {k: (v['achieved'], v['avg_launch_us']) for k, v in (r.get('hbm') or {}).items()})
print('cpu', d.get('cpu_baseline', {}).get('value'))
PY
