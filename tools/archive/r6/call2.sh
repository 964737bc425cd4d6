#!/bin/bash
# round 6, GPU call 2: (1) DP replay checks (advisor high / medium), (2) dead-workgroup cost of capacity-sized dyn launches, (3) bf16x3 NT per
# shape class inside the step, un-instrumented, three interleaved rounds
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
for m in "1 native" "flip native" "flip torch" "0 native"; do
  echo "== dp_replay_main $m" >> gpurun_out/r06b_dp.txt
  timeout 300 python tests/dp_replay_main.py $m 2>gpurun_out/r06b_dp_err.txt | python -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]) if l else {}
print({k:v for k,v in d.items() if k!='steps'})
for s in d.get('steps',[]): print('   ',s)" >> gpurun_out/r06b_dp.txt 2>&1
  tail -3 gpurun_out/r06b_dp_err.txt | grep -v amdgpu.ids >> gpurun_out/r06b_dp.txt
done
cat gpurun_out/r06b_dp.txt | cut -c1-400
for r in 0 1; do echo "NNR_BX3=$r"; NNR_BX3=$r timeout 200 python tools/gemm_dyn.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06b_dyn.txt
cat gpurun_out/r06b_dyn.txt
rm -f gpurun_out/r06b_classes.txt
ab() {
  echo -n "$1 : " >> gpurun_out/r06b_classes.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06b_classes.txt 2>&1
}
for i in 1 2 3; do
  ab "NNR_BX3=0"
  ab "NNR_BX3=1 NNR_BX3_CLASSES=dx"
  ab "NNR_BX3=1 NNR_BX3_CLASSES=dx,proj"
  ab "NNR_BX3=1 NNR_BX3_CLASSES=dx,gate"
  ab "NNR_BX3=1 NNR_BX3_CLASSES=dx,sue"
  ab "NNR_BX3=1 NNR_BX3_CLASSES=dx,proj,gate,sue"
done
cat gpurun_out/r06b_classes.txt
