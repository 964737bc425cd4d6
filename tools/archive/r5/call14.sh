#!/bin/bash
# round 5, GPU call 14: shader clock and socket power while the headline step runs -- fp32 path vs NNR_BX3=1 (is the bf16x3 path's bimodal step time a power / clock state?)
mkdir -p gpurun_out
OUT=gpurun_out/r05p_clocks.txt
rm -f $OUT
sample() {   # $1 = tag: one line per ~0.25 s: sclk, mclk, power
  while [ -f /tmp/sampling ]; do
    s=$(cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | grep '\*' | head -1 | tr -d '\n')
    p=$(cat /sys/class/drm/card*/device/hwmon/hwmon*/power1_average 2>/dev/null | head -1)
    f=$(cat /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input 2>/dev/null | head -1)
    echo "$1 sclk[$s] freq1 $f power_uW $p" >> $OUT
    sleep 0.25
  done
}
rocm-smi --showclocks --showpower 2>&1 | head -30 >> $OUT
for cfg in "NNR_BX3=0" "NNR_BX3=1" "NNR_BX3=0" "NNR_BX3=1"; do
  touch /tmp/sampling
  sample "$cfg" &
  SP=$!
  env $cfg timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 60 --sustained_seconds 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RESULT', d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> $OUT
  rm -f /tmp/sampling
  wait $SP
done
touch /tmp/sampling
sample "TN_MICRO_v3" &
SP=$!
for i in 1 2 3 4 5 6; do timeout 60 tools/micro/bf16x3_tn 400 400 450560 64 10 | grep variant | cut -c1-200 >> $OUT; done
for i in 1 2 3; do timeout 60 tools/micro/bf16x3_gemm 450560 400 400 2>&1 | grep -E "bf16x3|native" | cut -c1-160 >> $OUT; done
rm -f /tmp/sampling
wait $SP
grep -c . $OUT
python3 - <<'PY'
import re, collections
rows = collections.defaultdict(list)
for l in open('gpurun_out/r05p_clocks.txt'):
    m = re.match(r'(\S+) sclk\[(.*?)\] freq1 (\S*) power_uW (\S*)', l)
    if m:
        rows[m.group(1)].append((m.group(2), m.group(3), m.group(4)))
    elif l.startswith('RESULT') or 'variant' in l or 'native' in l or 'bf16x3' in l:
        print(l.strip()[:200])
for k, v in rows.items():
    f = [int(x[1]) / 1e6 for x in v if x[1].isdigit()]
    p = [int(x[2]) / 1e6 for x in v if x[2].isdigit()]
    print(k, 'samples', len(v), 'freq MHz min/median/max', (min(f), sorted(f)[len(f)//2], max(f)) if f else None, 'power W min/median/max', (min(p), sorted(p)[len(p)//2], max(p)) if p else None, 'sclk states', collections.Counter(x[0] for x in v).most_common(4))
PY
