#!/bin/bash
# round 6, GPU call 10: bf16x3 tile shapes alone at the step's shapes; the row threshold of the bf16x3 path on the small configs; tile 51 in the step
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
for t in 50 51 52; do echo "NNR_BX3_TILE=$t"; NNR_BX3_TILE=$t timeout 300 python tools/dyn_pair_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06j_tiles.txt
rm -f gpurun_out/r06j_ab.txt
ab() {
  echo -n "$1 $2: " >> gpurun_out/r06j_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], 'dominant', d['roofline']['family'], d['roofline']['avg_launch_us'], d['roofline']['step']['frac'])" >> gpurun_out/r06j_ab.txt 2>&1
}
for i in 1 2; do
  for c in "--config mhsa" "--batch_size 8" "--batch_size 16" "--batch_size 32"; do
    ab "NNR_BX3=0" "$c"; ab "NNR_BX3_MIN_ROWS=2048" "$c"; ab "NNR_BX3_MIN_ROWS=100000" "$c"
  done
  ab "NNR_BX3_TILE=50" ""; ab "NNR_BX3_TILE=51" ""; ab "NNR_BX3_MIN_ROWS=100000" ""
done
cat gpurun_out/r06j_ab.txt
