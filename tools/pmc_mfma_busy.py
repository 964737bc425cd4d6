#!/usr/bin/env python3
"""Matrix-pipe busy fraction per kernel from ONE rocprofv3 counter pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -- python3 bench.py --config mhsa ...
busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)   (GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1.0 = every
SIMD's matrix pipe busy every cycle = 157 TFLOP/s fp32; DESIGN.md section 4, profiles/r02_gemm_pmc_mfma_busy.txt).
Writes profiles-style JSON with the build id bench.py checks before quoting it (nnr_amd.profile.mfma_busy).
Usage: python tools/pmc_mfma_busy.py <dir with *counter_collection.csv> <out.json>"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnr_amd import _lib      # noqa: E402  (build_id only: no GPU call)


def short(n):
    return re.sub(r'^void ', '', n).replace('(anonymous namespace)::', '').split('(')[0]


agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            a = agg[short(r['Kernel_Name'])][r['Counter_Name']]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
out = []
for k, cs in agg.items():
    if 'GRBM_GUI_ACTIVE' in cs and 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and cs['GRBM_GUI_ACTIVE'][0] > 0:
        busy = cs['SQ_VALU_MFMA_BUSY_CYCLES'][0] / (cs['GRBM_GUI_ACTIVE'][0] / 8 * 1024)
        out.append(dict(kernel=k, launches=cs['GRBM_GUI_ACTIVE'][1], mfma_busy=round(busy, 4),
                        mfma_busy_cycles_per_launch=round(cs['SQ_VALU_MFMA_BUSY_CYCLES'][0] / cs['SQ_VALU_MFMA_BUSY_CYCLES'][1]),
                        gui_active_per_launch=round(cs['GRBM_GUI_ACTIVE'][0] / cs['GRBM_GUI_ACTIVE'][1])))
out.sort(key=lambda o: -o['mfma_busy_cycles_per_launch'] * o['launches'])
json.dump(dict(build_id=_lib.build_id(), note='SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) per kernel, one rocprofv3 --pmc pass '
               '(dispatches serialised by the profiler: solo figures)', kernels=out), open(sys.argv[2], 'w'), indent=1)
for o in out[:16]:
    print('%-52s %4d launches  mfma_busy %.3f' % (o['kernel'][:52], o['launches'], o['mfma_busy']))
