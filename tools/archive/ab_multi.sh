#!/bin/bash
# usage: ab_multi.sh VAR "v1 v2 v3" [bench args...]   -- two interleaved rounds of bench.py under each value of an environment knob
VAR=$1; VALS=$2; shift 2
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
for r in 1 2; do for v in $VALS; do echo -n "$VAR=$v  "; env $VAR=$v python bench.py --no_cpu_baseline "$@" 2>/dev/null | python -c "$P"; done; done
