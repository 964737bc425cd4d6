"""Timeline analysis of a rocprofv3 kernel trace (kernel_trace.csv): per-step GPU busy / idle time, per-queue busy
time, overlap, and the kernels on the critical stream.  Usage: python tools/timeline.py trace.csv [steps_to_skip]"""
import csv, sys, collections, re

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'), r.get('Stream_Id', '0')))
rows.sort()
# steps are delimited by the fused optimizer kernel
ends = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
print('kernels %d, optimizer steps %d' % (len(rows), len(ends)))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) // 2
a, b = ends[skip] + 1, ends[-1] + 1
seg = rows[a:b]
nstep = len(ends) - 1 - skip
t0, t1 = seg[0][0], max(r[1] for r in seg)
print('window: %d steps, %.3f ms/step, %d kernels/step' % (nstep, (t1 - t0) / 1e6 / nstep, len(seg) / nstep))
# union busy
busy, cur_s, cur_e = 0, None, None
for s, e, *_ in seg:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('GPU busy (union) %.3f ms/step, idle %.3f ms/step' % (busy / 1e6 / nstep, (t1 - t0 - busy) / 1e6 / nstep))
tot = sum(e - s for s, e, *_ in seg)
print('sum of kernel durations %.3f ms/step (overlap %.3f)' % (tot / 1e6 / nstep, (tot - busy) / 1e6 / nstep))
byq = collections.defaultdict(lambda: [0, 0])
for s, e, n, q, st in seg:
    byq[(q, st)][0] += e - s
    byq[(q, st)][1] += 1
for k, v in sorted(byq.items(), key=lambda kv: -kv[1][0]):
    print('  queue/stream %s: %.3f ms/step, %d kernels/step' % (k, v[0] / 1e6 / nstep, v[1] / nstep))
# gap histogram (idle gaps between union intervals)
gaps = []
cur_e = None
for s, e, *_ in seg:
    if cur_e is not None and s > cur_e:
        gaps.append(s - cur_e)
    cur_e = e if cur_e is None else max(cur_e, e)
gaps.sort()
if gaps:
    print('idle gaps: %d/step, median %.1f us, p90 %.1f us, max %.1f us' % (len(gaps) / nstep, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * .9)] / 1e3, gaps[-1] / 1e3))
# time by kernel name where ONLY that kernel is running (exclusive) vs total
name_t = collections.defaultdict(lambda: [0, 0])
for s, e, n, *_ in seg:
    short = re.sub(r'^void ', '', n).replace('(anonymous namespace)::', '').split('(')[0][:60]
    name_t[short][0] += e - s
    name_t[short][1] += 1
print('top kernels by summed duration:')
for k, v in sorted(name_t.items(), key=lambda kv: -kv[1][0])[:25]:
    print('  %-62s %8.3f ms/step  %6.1f launches/step  avg %7.1f us' % (k, v[0] / 1e6 / nstep, v[1] / nstep, v[0] / 1e3 / v[1]))
for key in sorted(byq, key=lambda k: -byq[k][0]):
    nt = collections.defaultdict(lambda: [0, 0])
    for s, e, n, q, st in seg:
        if (q, st) == key:
            short = re.sub(r'^void ', '', n).replace('(anonymous namespace)::', '').split('(')[0][:60]
            nt[short][0] += e - s
            nt[short][1] += 1
    print('stream %s:' % (key,))
    for k, v in sorted(nt.items(), key=lambda kv: -kv[1][0])[:14]:
        print('  %-62s %8.3f ms/step  %6.1f launches/step' % (k, v[0] / 1e6 / nstep, v[1] / nstep))

# ---- when is only ONE of the streams busy?  (side-only time = the main stream waiting at a join, or idle)
def union(iv):
    iv = sorted(iv); out = []
    for s_, e_ in iv:
        if out and s_ <= out[-1][1]: out[-1][1] = max(out[-1][1], e_)
        else: out.append([s_, e_])
    return out
def total(u): return sum(e_ - s_ for s_, e_ in u)
def inter(a, b):
    i = j = 0; t = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo: t += hi - lo
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return t
keys = sorted(byq, key=lambda k: -byq[k][0])
if len(keys) >= 2:
    um = union([(s_, e_) for s_, e_, n_, q_, st_ in seg if (q_, st_) == keys[0]])
    us = union([(s_, e_) for s_, e_, n_, q_, st_ in seg if (q_, st_) == keys[1]])
    both = inter(um, us)
    print('main busy %.3f, side busy %.3f, both %.3f, main-only %.3f, side-only %.3f ms/step' %
          (total(um) / 1e6 / nstep, total(us) / 1e6 / nstep, both / 1e6 / nstep, (total(um) - both) / 1e6 / nstep, (total(us) - both) / 1e6 / nstep))
    # side-only stretches longer than 50 us: which kernels run there
    gaps_m = []
    for i in range(len(um) - 1):
        gaps_m.append((um[i][1], um[i + 1][0]))
    big = [(a_, b_) for a_, b_ in gaps_m if b_ - a_ > 50000]
    agg = collections.defaultdict(float)
    for s_, e_, n_, q_, st_ in seg:
        if (q_, st_) != keys[1]: continue
        for a_, b_ in big:
            lo, hi = max(s_, a_), min(e_, b_)
            if hi > lo:
                short = re.sub(r'^void ', '', n_).replace('(anonymous namespace)::', '').split('(')[0][:50]
                agg[short] += hi - lo
    print('main-stream gaps > 50 us: %.3f ms/step in %d gaps/step; side kernels running inside them:' % (sum(b_ - a_ for a_, b_ in big) / 1e6 / nstep, len(big) / nstep))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:8]:
        print('   %-52s %.3f ms/step' % (k, v / 1e6 / nstep))
