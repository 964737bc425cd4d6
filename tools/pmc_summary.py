#!/usr/bin/env python3
"""Per-kernel averages of the counters of one or more rocprofv3 --pmc passes (counter_collection.csv files found under the
given directories).  Usage: pmc_summary.py dir [dir ...]"""
import collections, csv, glob, os, re, sys


def short(n):
    n = re.sub(r'^void ', '', n).replace('(anonymous namespace)::', '')
    return n.split('(')[0][:70]


agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for root in sys.argv[1:]:
    for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                a = agg[short(r['Kernel_Name'])][r['Counter_Name']]
                a[0] += float(r['Counter_Value'])
                a[1] += 1
for k, cs in sorted(agg.items()):
    v = {c: s / max(1, n) for c, (s, n) in cs.items()}
    line = '%-72s' % k
    if 'GRBM_GUI_ACTIVE' in v and 'SQ_VALU_MFMA_BUSY_CYCLES' in v and v['GRBM_GUI_ACTIVE'] > 0:
        line += ' mfma_busy %.3f' % (v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024))
    if 'SQ_WAVE_CYCLES' in v and v['SQ_WAVE_CYCLES'] > 0:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC', 'SQ_INST_CYCLES_SALU'):
            if c in v:
                line += ' %s %.3f' % (c.replace('SQ_', '').lower(), v[c] / v['SQ_WAVE_CYCLES'])
    line += ' | ' + ' '.join('%s=%.4g' % (c, x) for c, x in sorted(v.items()))
    print(line)
