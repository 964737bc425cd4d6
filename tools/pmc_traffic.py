#!/usr/bin/env python3
"""Aggregate the two rocprofv3 PMC passes of the bench (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, each with --kernel-trace,
--output-format csv) into per-kernel HBM traffic per launch, corrected as MI355X_MICROARCH.md (HBM / rocprofv3) prescribes:
FETCH_SIZE (KB) under-reports wide coalesced reads on gfx950 by exactly 2x -> doubled; WRITE_SIZE (KB) is exact.
Usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnr_amd import _lib      # noqa: E402  (build_id only: hashes of the kernel sources / the built library, no GPU call)


def short(n):
    return re.sub(r'^void ', '', n).replace('(anonymous namespace)::', '').split('(')[0]


BY_GRID = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))      # counter -> (kernel, slot in the step) -> [sum, launches]
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 16      # optimizer steps of the profiled command (--steps 12 --warmup 4)


def load(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    rows = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] == counter:
                a = agg[short(r['Kernel_Name'])]
                a[0] += float(r['Counter_Value'])
                a[1] += 1
                rows[short(r['Kernel_Name'])].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    # the launches of one kernel come in the same order every step (one launch shape per slot): average per slot
    for k, v in rows.items():
        if len(v) % STEPS == 0:
            per = len(v) // STEPS
            for i, (_, val) in enumerate(sorted(v)):
                g = BY_GRID[counter][(k, i % per)]
                g[0] += val
                g[1] += 1
    return agg


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out = []
for k in sorted(fetch, key=lambda k: -(2 * fetch[k][0] + write.get(k, [0, 0])[0])):
    fk, n = fetch[k]
    wk = write.get(k, [0.0, 0])[0]
    out.append(dict(kernel=k, launches=n, fetch_bytes_per_launch=round(2 * fk * 1024 / n), write_bytes_per_launch=round(wk * 1024 / max(1, write.get(k, [0, 1])[1])),
                    hbm_bytes_per_launch=round((2 * fk + wk) * 1024 / n)))
shapes = []
for (k, grid), (fk, n) in sorted(BY_GRID['FETCH_SIZE'].items(), key=lambda kv: -kv[1][0]):
    if 'gemm_tn' in k:       # the weight-gradient family: one row per launch shape (grid size = threads of the launch)
        wk, wn = BY_GRID['WRITE_SIZE'].get((k, grid), [0.0, 1])
        shapes.append(dict(kernel=k, slot_in_step=grid, launches=n, fetch_bytes_per_launch=round(2 * fk * 1024 / n),
                           write_bytes_per_launch=round(wk * 1024 / max(1, wn))))
json.dump(dict(build_id=_lib.build_id(), weight_gradient_launch_shapes=shapes, note='FETCH_SIZE x2 (gfx950 caveat) + WRITE_SIZE, bytes per launch averaged over the launches of one bench run', kernels=out),
          open(sys.argv[3], 'w'), indent=1)
for o in shapes:
    print('  %-40s slot %2s  %3d launches  fetch %8.1f MB  write %7.1f MB' % (o['kernel'][:40], o['slot_in_step'], o['launches'], o['fetch_bytes_per_launch'] / 1e6, o['write_bytes_per_launch'] / 1e6))
for o in out[:12]:
    print('%-44s %4d launches  fetch %8.1f MB  write %8.1f MB per launch' % (o['kernel'][:44], o['launches'], o['fetch_bytes_per_launch'] / 1e6, o['write_bytes_per_launch'] / 1e6))
