#!/bin/bash
# round 6, GPU call 1: where does the bf16x3 NT gain go?  Phase tables of the replayed batch-64 step with NNR_BX3=0/1 (same box),
# five interleaved un-instrumented bench pairs, and a quick sanity run of the suite's headline tests on this box.
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
for r in 0 1; do
  NNR_BX3=$r timeout 400 python tools/phase_table.py --json gpurun_out/r06a_phase_bx3_$r.json 2>&1 | grep -v amdgpu.ids > gpurun_out/r06a_phase_bx3_$r.txt
done
python tools/phase_table.py --diff gpurun_out/r06a_phase_bx3_0.json gpurun_out/r06a_phase_bx3_1.json > gpurun_out/r06a_phase_diff.md 2>&1
cat gpurun_out/r06a_phase_diff.md
rm -f gpurun_out/r06a_pairs.txt
ab() {
  echo -n "$1 : " >> gpurun_out/r06a_pairs.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06a_pairs.txt 2>&1
}
for i in 1 2 3 4 5; do ab "NNR_BX3=0"; ab "NNR_BX3=1"; done
cat gpurun_out/r06a_pairs.txt
