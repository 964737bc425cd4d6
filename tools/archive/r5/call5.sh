#!/bin/bash
# round 5, GPU call 5: MHSA news encoder over packed token rows (NNR_MHSA_PACKED): unit + parity tests, config-2 A/B
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_hip_ops_gpu.py tests/test_hip_edge_gpu.py tests/test_hip_model_gpu.py tests/test_hip_dropout_gpu.py tests/test_hip_tape_gpu.py tests/test_hip_eval_gpu.py tests/test_hip_layers_gpu.py -q -x -k "mhsa or MHSA or pool or golden or fixture or eval or attention or layers" 2>&1 | tail -25) > gpurun_out/r05g_tests.log
tail -6 gpurun_out/r05g_tests.log
(timeout 600 python -m pytest tests/test_hip_headline_gpu.py -q -x -k "mhsa" 2>&1 | tail -8) >> gpurun_out/r05g_tests.log
tail -3 gpurun_out/r05g_tests.log
for i in 1 2; do
  for v in 0 1; do
    echo "NNR_MHSA_PACKED=$v --config mhsa" >> gpurun_out/r05g_ab.txt
    NNR_MHSA_PACKED=$v timeout 300 python bench.py --config mhsa --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], d['roofline']['step'], {k: (v['avg_launch_us'], v['hbm_gb_s']) for k, v in (d['roofline'].get('mhsa') or {}).items()})" >> gpurun_out/r05g_ab.txt
  done
done
cat gpurun_out/r05g_ab.txt
