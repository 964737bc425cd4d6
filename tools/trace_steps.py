#!/usr/bin/env python3
"""Per-step timeline from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py (no per-call HIP events: the profiler's
timestamps are taken by the hardware queue, the step runs as it does un-instrumented apart from the tracing overhead).  A step ends with
adam_kernel; for the LAST `--steps` steps: per (kernel, grid size) the median start offset inside the step, the median duration and the
summed duration, ordered by start -- i.e. tools/tape_timeline.py's table without its ~10 % stretch.

    python tools/trace_steps.py <kernel_trace.csv> [--steps 8] [--min_us 15]"""
import argparse
import csv
import re
import statistics

ap = argparse.ArgumentParser()
ap.add_argument('csv')
ap.add_argument('--steps', type=int, default=8)
ap.add_argument('--min_us', type=float, default=15.0)
a = ap.parse_args()


def short(name):
    n = re.sub(r'^void\s+', '', name.strip()).replace('(anonymous namespace)::', '')
    depth, out = 0, []
    for ch in n:
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            break
        out.append(ch)
    return ''.join(out).strip()


rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        gx = r.get('Grid_Size_X') or r.get('Grid_Size') or '0'
        wx = r.get('Workgroup_Size_X') or r.get('Workgroup_Size') or '1'
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), int(gx) // max(1, int(wx)), r.get('Stream_Id') or r.get('Queue_Id') or ''))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2].startswith('adam_kernel')]
if len(ends) < a.steps + 1:
    raise SystemExit('only %d optimizer steps in the trace' % len(ends))
steps = []
for k in range(len(ends) - a.steps, len(ends)):
    lo, hi = ends[k - 1] + 1, ends[k]
    seg = rows[lo:hi + 1]
    t0 = seg[0][0]
    steps.append((seg, t0, rows[hi][1] - t0))
print('steps %d; span first kernel start .. adam end: median %.3f ms (min %.3f, max %.3f)' % (
    len(steps), statistics.median(s[2] for s in steps) / 1e6, min(s[2] for s in steps) / 1e6, max(s[2] for s in steps) / 1e6))
# align the k-th occurrence of (kernel, workgroups) across steps
agg = {}
for seg, t0, _ in steps:
    seen = {}
    for s, e, name, wgs, q in seg:
        key0 = (name, wgs)
        i = seen.get(key0, 0)
        seen[key0] = i + 1
        agg.setdefault((name, wgs, i), []).append((s - t0, e - s, q))
table = []
for (name, wgs, i), v in agg.items():
    if len(v) < len(steps) // 2:
        continue
    table.append((statistics.median(x[0] for x in v) / 1e3, statistics.median(x[1] for x in v) / 1e3, name, wgs, i, v[0][2]))
table.sort()
tot = 0.0
for st, du, name, wgs, i, q in table:
    tot += du
    if du >= a.min_us:
        print('%9.1f %8.1f  q%-3s wgs %-7d %s' % (st, du, q[-3:], wgs, name[:80]))
print('sum of median durations %.3f ms over %d kernels per step' % (tot / 1e3, len(table)))
