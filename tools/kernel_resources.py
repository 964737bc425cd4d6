#!/usr/bin/env python3
"""Register / LDS / scratch use of the compiled kernels (the code object's metadata notes), e.g. to check a new template
instance for spills without a GPU.  Usage: python tools/kernel_resources.py [pattern] [object file, default: every csrc/build/*.o]"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else ''
objs = sys.argv[2:] or sorted(glob.glob(os.path.join(ROOT, 'nnr_amd', 'csrc', 'build', '*.o')))
for o in objs:
    # the device code object sits in the host object's .hip_fatbin section as an offload bundle: extract, then unbundle
    fat, tmp = '/tmp/_kr_%s.fatbin' % os.path.basename(o), '/tmp/_kr_%s.co' % os.path.basename(o)
    for f in (fat, tmp):
        if os.path.exists(f):
            os.remove(f)
    subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', o, fat], check=False, capture_output=True)
    if not os.path.exists(fat) or os.path.getsize(fat) == 0:
        continue
    subprocess.run(['/opt/rocm/lib/llvm/bin/clang-offload-bundler', '--unbundle', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                    '--input=' + fat, '--output=' + tmp], check=False, capture_output=True)
    if not os.path.exists(tmp) or os.path.getsize(tmp) == 0:
        continue
    txt = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', tmp], capture_output=True, text=True).stdout
    for blk in txt.split('  - .agpr_count')[1:]:
        name = re.search(r'\.name:\s+(\S+)', blk)
        if not name:
            continue
        dem = subprocess.run(['c++filt', name.group(1)], capture_output=True, text=True).stdout.strip()
        if pat and pat not in dem:
            continue
        g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, blk) or [None, '?'])[1]
        agpr = re.match(r':\s+(\d+)', blk)
        print('%-70s vgpr %s agpr %s sgpr %s lds %s scratch %s spill_v %s' % (re.sub(r'^void ', '', dem).replace('(anonymous namespace)::', '').split('(')[0][:70], g('vgpr_count'), agpr.group(1) if agpr else '?', g('sgpr_count'),
              g('group_segment_fixed_size'), g('private_segment_fixed_size'), g('vgpr_spill_count')))
