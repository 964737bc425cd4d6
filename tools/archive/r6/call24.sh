#!/bin/bash
# round 6, GPU call 24: tile of the d c_n product in front of the backward recurrence (74 KB tile waits for the leaf stream's weight-gradient workgroups to drain)
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
rm -f gpurun_out/r06x_ab.txt
ab() {
  echo -n "$1 : " >> gpurun_out/r06x_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']; print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], {k: (v['ms'], v['launches']) for k, v in f.items() if 'gemm_nt_64x80' in k or 'gemm_nn_64x80' in k})" >> gpurun_out/r06x_ab.txt 2>&1
}
for i in 1 2 3; do ab "NNR_DCN_TILE=0"; ab "NNR_DCN_TILE=2"; ab "NNR_DCN_TILE=4"; done
cat gpurun_out/r06x_ab.txt
