#!/bin/bash
# usage: ab_env.sh VAR val0 val1 [bench args...]   -- interleaved same-box A/B of bench.py under an environment knob
VAR=$1; A=$2; B=$3; shift 3
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
for v in $A $B $A $B; do echo -n "$VAR=$v  "; env $VAR=$v python bench.py --no_cpu_baseline "$@" 2>/dev/null | python -c "$P"; done
