#!/bin/bash
# round 6, GPU call 31: MHSA step with the 'dx' class on the bf16x3 kernel (default now) vs none; then the full GPU suite on the new build
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
rm -f gpurun_out/r06E_ab.txt
ab() {
  echo -n "$1 : " >> gpurun_out/r06E_ab.txt
  env $1 timeout 300 python bench.py --config mhsa --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], d['config']['matrix_path']['nt_weight_gemms'][:60], d['config']['matrix_path'].get('launch_classes'))" >> gpurun_out/r06E_ab.txt 2>&1
}
for i in 1 2 3; do ab "NNR_BX3_MHSA_CLASSES="; ab "NNR_BX3_MHSA_CLASSES=dx"; done
cat gpurun_out/r06E_ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06E_tests.log 2>&1
tail -5 gpurun_out/r06E_tests.log
