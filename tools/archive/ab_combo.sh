#!/bin/bash
# usage: ab_combo.sh "A=1 B=2" "A=0 B=2" ... -- two interleaved rounds of bench.py (batch 64) under each environment combination
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
for r in 1 2; do for c in "$@"; do echo -n "$c  "; env $c python bench.py --no_cpu_baseline $BENCH_ARGS 2>/dev/null | python -c "$P"; done; done
