#!/bin/bash
# round 6, GPU call 29: one-launch expand / fusion rows in the MHSA step: tests + configs[1] A/B (NNR_EXPAND_OLD=1 = the five add2d launches per direction)
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
timeout 1500 python -m pytest tests -x -q -m gpu -k "mhsa or MHSA or embed_gather_scatter or cabi or golden or elementwise" > gpurun_out/r06C_tests.log 2>&1
tail -4 gpurun_out/r06C_tests.log
rm -f gpurun_out/r06C_ab.txt
ab() {
  echo -n "$1 : " >> gpurun_out/r06C_ab.txt
  env $1 timeout 300 python bench.py --config mhsa --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06C_ab.txt 2>&1
}
for i in 1 2 3 4; do ab "NNR_EXPAND_OLD=1"; ab "NNR_EXPAND_OLD=0"; done
cat gpurun_out/r06C_ab.txt
