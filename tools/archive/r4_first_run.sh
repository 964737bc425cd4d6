#!/bin/bash
# Round 4: why is the FIRST bench process on a fresh box slower over its 20-step window than the second (10.9-11.2 vs 10.5-10.6 ms)?
# Per-step marks of the driver's own command, first process vs later ones, and with a longer warm-up.
O=gpurun_out/r04i; mkdir -p $O
export NNR_BENCH_STEP_MARKS=1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/run1.json 2> $O/run1.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/run2.json 2> $O/run2.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline --no_isolated --sustained_seconds 0 > $O/run3.json 2> $O/run3.err
python3 bench.py --gpus 1 --steps 20 --warmup 40 --no_cpu_baseline --no_isolated --sustained_seconds 0 > $O/run4_w40.json 2> $O/run4_w40.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --roofline_every 100 --no_cpu_baseline --no_isolated --sustained_seconds 0 > $O/run5_noinstr.json 2> $O/run5_noinstr.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --prebuilt --no_cpu_baseline --no_isolated --sustained_seconds 0 > $O/run6_prebuilt.json 2> $O/run6_prebuilt.err
for f in run1 run2 run3 run4_w40 run5_noinstr run6_prebuilt; do echo "== $f"; grep "step marks" $O/$f.err; python3 -c "
import json,sys
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step'))"; done
