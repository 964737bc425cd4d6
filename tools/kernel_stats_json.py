#!/usr/bin/env python3
"""profiles/kernel_stats.json from two `rocprofv3 --kernel-trace --stats` tables of the headline command: inside the step and with every HIP
stream collapsed into one (NNR_ONE_STREAM=1), stamped with the build id of the library that ran -- bench.py's `roofline.rocprof` quotes it only
when the running build is the same (nnr_amd/profile.py: rocprof_block).

    python tools/kernel_stats_json.py <in_step kernel_stats.csv> <one_stream kernel_stats.csv> <out.json>"""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    """'void (anonymous namespace)::gemm_tn_pipe2_kernel<1, 13, 3, 2>(nnr_gemm_args)' -> 'gemm_tn_pipe2_kernel<1, 13, 3, 2>'"""
    n = re.sub(r'^void\s+', '', name.strip())
    n = n.replace('(anonymous namespace)::', '')
    depth, out = 0, []
    for ch in n:                          # cut the parameter list: the first '(' outside template brackets
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            break
        out.append(ch)
    return ''.join(out).strip()


def table(path):
    res = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row['Name'])
            calls, tot = int(row['Calls']), float(row['TotalDurationNs'])
            e = res.setdefault(k, {'calls': 0, 'total_us': 0.0})
            e['calls'] += calls
            e['total_us'] += tot / 1000.0
    for e in res.values():
        e['avg_us'] = round(e['total_us'] / max(1, e['calls']), 2)
        e['total_us'] = round(e['total_us'], 1)
    return res


if __name__ == '__main__':
    from nnr_amd import _lib
    ins, solo, out = sys.argv[1:4]
    d = {'build_id': _lib.build_id(), 'in_step': table(ins), 'solo': table(solo),
         'sources': {'in_step': os.path.basename(ins), 'solo': os.path.basename(solo)}}
    json.dump(d, open(out, 'w'), indent=1, sort_keys=True)
    top = sorted(d['in_step'].items(), key=lambda kv: -kv[1]['total_us'])[:12]
    for k, v in top:
        s = d['solo'].get(k)
        print('%-60s in-step %8.1f us x %4d   solo %s' % (k[:60], v['avg_us'], v['calls'], ('%8.1f us' % s['avg_us']) if s else '-'))
