#!/bin/bash
# same-box A/B of scheduling knobs under the native replay: tools/ab_r3.sh "VAR=a VAR=b ..." (each entry: env assignments joined by ',')
for rep in 1 2; do
for cfg in "$@"; do
  env $(echo $cfg | tr ',' ' ') python bench.py --no_cpu_baseline --no_isolated --sustained_seconds 2 > /tmp/ab.json 2>/dev/null
  python - "$cfg" <<'PY'
import json,sys
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print('%-40s %8.1f imp/s  %7.3f ms (20 steps)  %7.3f ms sustained' % (sys.argv[1], d['value'], d['ms_per_step'], d['sustained']['ms_per_step']))
PY
done
done
