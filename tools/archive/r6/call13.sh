#!/bin/bash
# round 6, GPU call 13: the rest of the GPU suite after the tile-1 fix (continue where call 12 stopped), sue_intra XCD mapping (tests + in-step)
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
(timeout 2000 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -25) > gpurun_out/r06m_tests.log
tail -8 gpurun_out/r06m_tests.log | cut -c1-300
rm -f gpurun_out/r06m_ab.txt
ab() {
  echo -n "$1 $2: " >> gpurun_out/r06m_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['roofline']['hbm']; print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], 'sue_intra_fwd', h.get('sue_intra_fwd',{}).get('avg_launch_us'), 'bwd', h.get('sue_intra_bwd',{}).get('avg_launch_us'))" >> gpurun_out/r06m_ab.txt 2>&1
}
for i in 1 2 3; do ab "X=1" ""; done
cat gpurun_out/r06m_ab.txt
