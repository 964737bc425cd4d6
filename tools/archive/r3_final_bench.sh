#!/bin/bash
# Final bench lines of the round on the final build (profiles/pmc_traffic.json of the same build in place):
#   default headline line, then the per-GPU batch sweep (8 / 16 / 32 / 128) without the CPU leg.    tools/r3_final_bench.sh TAG
TAG=${1:-r03f}
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
for B in 8 16 32 128; do
  python bench.py --batch_size $B --no_cpu_baseline --no_isolated --sustained_seconds 1 --steps 40 --warmup 8 > gpurun_out/${TAG}_bench_b$B.json 2>> gpurun_out/${TAG}_bench.err
done
python bench.py --news_encoder MHSA --user_encoder MHSA --no_cpu_baseline --no_isolated --sustained_seconds 1 > gpurun_out/${TAG}_bench_mhsa.json 2>> gpurun_out/${TAG}_bench.err
python - <<PY
import json
for n in ['', '_b8', '_b16', '_b32', '_b128', '_mhsa']:
    try:
        d = json.load(open('gpurun_out/${TAG}_bench%s.json' % n))
        print(n or 'b64', d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step'), d['roofline'].get('traffic'))
    except Exception as e:
        print(n, 'FAILED', e)
PY
