for args in "--warmup 5" "--warmup 40" "--warmup 5 --roofline_every 1000" "--warmup 40 --roofline_every 1000" "--warmup 5 --steps 100" "--warmup 5" ; do
  python bench.py --no_cpu_baseline --no_isolated --sustained_seconds 1 $args > /tmp/w.json 2>/dev/null
  python - "$args" <<'PY'
import json,sys
d=json.loads(open('/tmp/w.json').read().strip().splitlines()[-1])
print('%-45s %7.3f ms (%d steps)  %7.3f ms sustained' % (sys.argv[1], d['ms_per_step'], d['steps'], d['sustained']['ms_per_step']))
PY
done
