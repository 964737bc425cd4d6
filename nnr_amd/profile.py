"""Live per-kernel-family timing for bench.py: HIP events recorded on the launch stream around each instrumented
launch (the kernels run on torch's current stream, so torch.cuda.Event sees them), plus the ALGORITHMIC FLOPs of the
launch (true dims, not the padded tile dims; data-dependent token counts are read back from device memory after the
timed region).  Families map 1:1 to kernel symbols, so the rocprofv3 --kernel-trace --stats averages under profiles/
can be compared directly."""
import os

import torch

_on = False
_records = []          # (family, start, end, flops_fn)

KERNEL_OF = {
    'gemm_nt_256x80': 'gemm_kernel<4, 5, false, false, 16>', 'gemm_nn_256x80': 'gemm_kernel<4, 5, false, true, 16>',
    'gemm_tn_256x80': 'gemm_kernel<4, 5, true, true, 16>', 'gemm_nt_64x80': 'gemm_kernel<1, 5, false, false, 16>',
    'gemm_nn_64x80': 'gemm_kernel<1, 5, false, true, 16>', 'gemm_tn_64x80': 'gemm_kernel<1, 5, true, true, 16>',
    'gemm_nt_128x208': 'gemm_kernel<2, 13, false, false, 16>', 'gemm_nt_128x80': 'gemm_kernel<2, 5, false, false, 16>', 'gemm_nt_128x80k32': 'gemm_kernel<2, 5, false, false, 32>',
    'gemm_nt_64x80k64': 'gemm_kernel<1, 5, false, false, 64>', 'gemm_nn_64x80k64': 'gemm_kernel<1, 5, false, true, 64>',
    'gemm_tn_64x80k64': 'gemm_kernel<1, 5, true, true, 64>',
    'gemm_nt_pipe128x80': 'gemm_nt_pipe_kernel<2, 5, 16, 3, 4, 0>', 'gemm_nt_pipe128x80s2': 'gemm_nt_pipe_kernel<2, 5, 16, 2, 5, 0>',
    'gemm_nt_bx3_128x80': 'gemm_nt_bx3_kernel<2, 5, 2>', 'gemm_nt_bx3_64x80': 'gemm_nt_bx3_kernel<1, 5, 2>',
    'gemm_nt_pipe2_128x80': 'gemm_nt_pipe2_kernel<2, 5, 3, 2>', 'gemm_nt_pipe2_128x64': 'gemm_nt_pipe2_kernel<2, 4, 3, 2>',
    'gemm_tn_pipe2_128x80': 'gemm_tn_pipe2_kernel<2, 5, 3, 3>', 'gemm_tn_pipe2_128x208': 'gemm_tn_pipe2_kernel<2, 13, 3, 2>', 'gemm_tn_pipe2_128x160': 'gemm_tn_pipe2_kernel<2, 10, 3, 2>', 'gemm_tn_pipe2_64x208': 'gemm_tn_pipe2_kernel<1, 13, 3, 2>',
    'gemm_tn_pipe128x80': 'gemm_tn_pipe_kernel<2, 5, 3, 3, 0>', 'gemm_tn_pipe128x208': 'gemm_tn_pipe_kernel<2, 13, 3, 2, 0>',
    'gemm_nn_128x80': 'gemm_kernel<2, 5, false, true, 16>', 'gemm_tn_128x80': 'gemm_kernel<2, 5, true, true, 16>', 'lstm_fwd': 'lstm_fwd_pair_kernel<13>', 'lstm_bwd': 'lstm_bwd_pair_kernel<13, true>' if os.environ.get('NNR_LSTM_BWD_BAL', '1') != '0' else 'lstm_bwd_pair_kernel<13, false>',
}


_enabled, _every = False, 1
TAPE_HOOK = [None]     # set by nnr_amd.tape while a step is being recorded: the next recorded call gets (family, flops_fn) as its tag
TAPE_RECORDS = []      # (family, flops_fn, ms) of timing replays, appended by the trainer (same role as _records for eager launches)


_eager = True


def enable(every=1, eager=True):
    """Start recording spans; with every > 1 only the steps announced by begin_step(i) with i % every == 0 are instrumented
    (two HIP events per launch cost ~5 % of a step when every launch of every step carries them).  eager=False: the steps are
    REPLAYED from a tape (nnr_amd.tape): no torch events around the Python-side launches -- begin_step() only says which steps to
    instrument and the trainer asks the tape for a timing replay (HIP events recorded natively around the same calls)."""
    global _on, _enabled, _every, _eager
    _enabled, _every, _eager = True, max(1, int(every)), bool(eager)
    _on = _eager
    _records.clear()
    TAPE_RECORDS.clear()


def begin_step(i):
    """Returns whether step i is an instrumented one."""
    global _on
    inst = _enabled and (i % _every == 0)
    _on = inst and _eager
    return inst


def disable():
    global _on, _enabled
    _on = _enabled = False


def active():
    return _on or TAPE_HOOK[0] is not None


class span:
    """with profile.span(family, flops_fn): <launch>"""

    def __init__(self, family, flops_fn):
        self.family, self.flops_fn = family, flops_fn

    def __enter__(self):
        if TAPE_HOOK[0] is not None:
            TAPE_HOOK[0](self.family, self.flops_fn)
        if _on:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if _on:
            self.e.record()
            _records.append((self.family, self.s, self.e, self.flops_fn))
        return False


def by_shape():
    """Diagnostic: time and achieved TFLOP/s per (family, shape tag)."""
    out = {}
    for family, fn, ms in _all_records():
        tag = getattr(fn, 'tag', '')
        d = out.setdefault((family, tag), dict(ms=0.0, flops=0.0, launches=0))
        d['ms'] += ms
        d['flops'] += float(fn())
        d['launches'] += 1
    return out


def _all_records():
    """(family, flops_fn, ms) of every instrumented launch: eager launches (torch events) and replayed ones (the tape's events)."""
    torch.cuda.synchronize()
    for family, s, e, fn in _records:
        yield family, fn, s.elapsed_time(e)
    for family, fn, ms in TAPE_RECORDS:
        yield family, fn, ms


def _bytes_of(fn):
    if hasattr(fn, 'op_bytes'):
        return float(fn.op_bytes)
    if hasattr(fn, 'bytes_fn'):
        return float(fn.bytes_fn())
    return None


def summary(hbm=False):
    """Per family: summed HIP-event time, algorithmic FLOPs and launches.  hbm=False: the MFMA-bound families (GEMMs, recurrences);
    hbm=True: the HBM-bound ones (ops._hbm_span: gathers / scatters / pools / element-wise / optimizer), with their algorithmic bytes."""
    fam = {}
    for family, fn, ms in _all_records():
        if bool(getattr(fn, 'hbm', False)) != bool(hbm):
            continue
        d = fam.setdefault(family, dict(ms=0.0, flops=0.0, launches=0, bytes=0.0, executed=0.0))
        d['ms'] += ms
        f = float(fn())
        d['flops'] += f
        d['executed'] += f / float(getattr(fn, 'scale', 1.0) or 1.0)
        d['launches'] += 1
        if hbm:
            d['bytes'] += _bytes_of(fn) or 0.0
    return fam


# HBM-bound families -> the kernels one call launches (names as tools/pmc_traffic.py shortens them; prefix match), kernels per call
HBM_KERNELS = {
    'embed_gather': (('embed_gather_kernel',), 1), 'embed_scatter': (('embed_scatter_kernel', 'embed_scatter_sorted_kernel'), 1),
    # (round 6: the packed token streams run pool_packed_kernel, only the user encoder's dense pools still run pool_kernel)
    'pool_fwd': (('pool_kernel<false', 'pool_packed_kernel<false'), 1), 'pool_bwd': (('pool_kernel<true', 'pool_packed_kernel<true'), 1), 'gate_bwd': (('gate_bwd_kernel',), 1),
    'gcn_aggregate_fwd': (('gcn_aggregate_kernel<0',), 1), 'gcn_aggregate_bwd': (('gcn_aggregate_kernel<1',), 1),
    'sue_intra_fwd': (('sue_intra_fwd_kernel',), 1), 'sue_intra_bwd': (('sue_intra_bwd_ds_kernel', 'sue_intra_bwd_dg_kernel'), 2),
    'clip_adam': (('adam_kernel',), 1), 'sumsq': (('sumsq_kernel',), 1),
}
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable by a float4 copy)


def pmc_traffic_family(family):
    """Counter HBM bytes per CALL of an HBM-bound family from the committed PMC passes (same build only), or None."""
    d = _pmc_file()
    spec = HBM_KERNELS.get(family)
    if d is None or spec is None:
        return None
    names, per_call = spec
    tot = launches = 0.0
    for k in d.get('kernels', []):
        if any(k['kernel'].startswith(n) for n in names):
            tot += k['hbm_bytes_per_launch'] * k['launches']
            launches += k['launches']
    return round(tot / (launches / per_call)) if launches else None


def mfma_busy(kernel_prefix):
    """Matrix-pipe busy fraction of the kernels whose name starts with `kernel_prefix` from the committed counter pass
    (profiles/pmc_mfma_busy.json, written by tools/pmc_mfma_busy.py from `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`
    of bench.py --config mhsa), quoted only when that file was collected on THIS build; else None."""
    import json
    import os
    from . import _lib
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'pmc_mfma_busy.json')
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None
    have, now = d.get('build_id') or {}, _lib.build_id()
    if not (have.get('src_sha256') == now['src_sha256'] or (have.get('lib_sha256') and have.get('lib_sha256') == now['lib_sha256'])):
        return None
    hits = [k for k in d.get('kernels', []) if k['kernel'].startswith(kernel_prefix)]
    if not hits:
        return None
    # several instantiations of a family (the news encoder's and the user encoder's attention core): cycles-weighted over all launches
    busy = sum(k['mfma_busy_cycles_per_launch'] * k['launches'] for k in hits)
    active = sum(k['gui_active_per_launch'] * k['launches'] for k in hits)
    return {'kernel': ' + '.join(k['kernel'] for k in hits), 'launches': sum(k['launches'] for k in hits),
            'mfma_busy': round(busy / (active / 8.0 * 1024.0), 4) if active else None}


def mhsa_roofline(peak_tflops):
    """bench.py --config mhsa, `roofline.mhsa`: the MFMA attention core (csrc/mhsa.hip) per direction -- algorithmic FLOPs of the
    Q K^T / P V contractions over live HIP-event time against the fp32 MFMA peak (= the arithmetic MFMA utilisation), its algorithmic
    bytes against the HBM peak (the kernel is HBM-bound: 32 x 32 x 20 attention matrices), and the matrix-pipe busy counter."""
    fam = summary()
    out = {}
    for k, kern in (('mhsa_fwd', 'mhsa_fwd'), ('mhsa_bwd', 'mhsa_bwd')):      # (prefixes: mhsa_bwd_kernel<..> and mhsa_bwd_persist_kernel<..>)
        v = fam.get(k)
        if not v or v['ms'] <= 0:
            continue
        tf = v['flops'] / (v['ms'] * 1e-3) / 1e12
        nbytes = sum((_bytes_of(fn) or 0.0) for f, fn, _ in _all_records() if f == k)
        gbs = nbytes / (v['ms'] * 1e-3) / 1e9
        busy = mfma_busy(kern)
        out[k] = {'kernel': kern + '*_kernel' if busy is None else busy['kernel'], 'launches': v['launches'], 'avg_launch_us': round(1000 * v['ms'] / v['launches'], 2),
                  'mfma_tflops': round(tf, 2), 'mfma_utilisation': round(tf / peak_tflops, 4),
                  'hbm_gb_s': round(gbs, 1), 'hbm_frac': round(gbs / PEAK_HBM_GBS, 4),
                  'mfma_busy_counter': None if busy is None else busy.get('mfma_busy'),
                  'mfma_busy_source': 'profiles/pmc_mfma_busy.json (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), same build only)'}
    return out or None


def hbm_roofline():
    """bench.py `roofline.hbm`: per HBM-bound family of the step, algorithmic bytes / live HIP-event time against the 8 TB/s peak,
    beside the counter bytes of the committed PMC passes.  The durations are IN-STEP (the launch shares the chip with up to three
    other streams), so `frac` is a lower bound of what the kernel reaches alone."""
    fam = summary(hbm=True)
    out = {}
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms']):
        if v['ms'] <= 0 or not v['launches']:
            continue
        gbs = v['bytes'] / (v['ms'] * 1e-3) / 1e9
        t = pmc_traffic_family(k)
        per = v['bytes'] / v['launches']
        out[k] = {'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 4), 'launches': v['launches'],
                  'avg_launch_us': round(1000 * v['ms'] / v['launches'], 2), 'algorithmic_bytes_per_launch': round(per),
                  'traffic': t, 'traffic_over_algorithmic': None if not t else round(t / per, 2)}
    return out or None


_PMC = {}


def _pmc_file():
    """profiles/pmc_traffic.json if it belongs to THIS build (its build_id equals the sources' or the loaded binary's hash), else
    None; the reason is kept for the bench line."""
    if 'data' not in _PMC:
        import json
        import os
        from . import _lib
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'pmc_traffic.json')
        _PMC['data'], _PMC['why'] = None, 'profiles/pmc_traffic.json missing or unreadable'
        try:
            d = json.load(open(path))
            have, now = d.get('build_id') or {}, _lib.build_id()
            if have.get('src_sha256') == now['src_sha256'] or (have.get('lib_sha256') and have.get('lib_sha256') == now['lib_sha256']):
                _PMC['data'], _PMC['why'] = d, 'build_id matches (%s)' % now['src_sha256']
            else:
                _PMC['why'] = 'profiles/pmc_traffic.json was collected on another build (%s, running %s): not quoted' % (have.get('src_sha256'), now['src_sha256'])
        except (OSError, ValueError, KeyError):
            pass
    return _PMC['data']


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command ON THIS BUILD (profiles/pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, corrected for gfx950 by tools/pmc_traffic.py), or None."""
    d = _pmc_file()
    if d is None:
        return None
    for k in d.get('kernels', []):
        if k['kernel'] == kernel:
            return k['hbm_bytes_per_launch']
    return None


def tn_operand_bytes():
    """Per weight-gradient (token-reduction GEMM) family: algorithmic operand bytes per launch = live reduction rows x (M + N) x 4,
    over the recorded launches -- the figure the PMC traffic of profiles/pmc_traffic.json is compared with."""
    out = {}
    for family, fn, _ms in _all_records():
        dims = getattr(fn, 'tn_dims', None)
        if dims:
            M, N, fs = dims
            d = out.setdefault(family, dict(bytes=0.0, launches=0))
            d['bytes'] += float(fn()) / (2.0 * M * N * fs) * (M + N) * 4.0
            d['launches'] += 1
    return out


def operand_bytes(family):
    """Algorithmic HBM bytes per launch of a GEMM family over the recorded launches: every operand read once, every output written
    once (fn.op_bytes, set by ops.gemm from the true -- device-side -- extents)."""
    tot, n = 0.0, 0
    for fam, fn, _ms in _all_records():
        if fam != family:
            continue
        b = fn.op_bytes if hasattr(fn, 'op_bytes') else (fn.bytes_fn() if hasattr(fn, 'bytes_fn') else None)
        if b is not None:
            tot += float(b)
            n += 1
    return (tot / n) if n else None


def weight_gradient_traffic():
    res = {}
    tot_op = tot_hbm = 0.0
    for family, d in tn_operand_bytes().items():
        hbm = pmc_traffic(KERNEL_OF.get(family, family))
        if hbm and d['launches']:
            op = d['bytes'] / d['launches']
            res[family] = {'operand_mb_per_launch': round(op / 1e6, 1), 'counter_mb_per_launch': round(hbm / 1e6, 1), 'ratio': round(hbm / op, 2)}
            tot_op += d['bytes']
            tot_hbm += hbm * d['launches']
    if tot_op:
        res['all'] = {'ratio': round(tot_hbm / tot_op, 2)}
    return res or None


def roofline(peak_tflops, sampled_steps=None, ms_per_step=None):
    fam = summary()
    if not fam:
        return None
    total_ms = sum(d['ms'] for d in fam.values())
    name = max(fam, key=lambda k: fam[k]['ms'])
    d = fam[name]
    ach = d['flops'] / (d['ms'] * 1e-3) / 1e12 if d['ms'] > 0 else 0.0
    return {
        'bound': 'mfma', 'achieved': round(ach, 3), 'peak': peak_tflops, 'unit': 'TFLOP/s', 'frac': round(ach / peak_tflops, 4),
        'traffic': pmc_traffic(KERNEL_OF.get(name, name)), 'traffic_source': 'profiles/pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes '
        'of this command (PMC cannot be collected inside the timed run), FETCH_SIZE x2 per the gfx950 note; quoted only when the file\'s build_id equals the running build: ' + _PMC.get('why', ''),
        'algorithmic_bytes_per_launch': (lambda b: None if b is None else round(b))(operand_bytes(name)),
        'traffic_over_algorithmic': (lambda t, b: None if not (t and b) else round(t / b, 2))(pmc_traffic(KERNEL_OF.get(name, name)), operand_bytes(name)),
        'kernel': KERNEL_OF.get(name, name), 'family': name, 'launches': d['launches'],
        'avg_launch_us': round(1000 * d['ms'] / max(1, d['launches']), 2),
        'share_of_instrumented_time': round(d['ms'] / total_ms, 3),
        'families': {k: {'ms': round(v['ms'], 3), 'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) if v['ms'] > 0 else 0.0,
                         'launches': v['launches']} for k, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms'])},
        # whole-step view: the launches of several HIP streams overlap, so per-kernel wall durations double-count the chip;
        # algorithmic FLOPs of all instrumented GEMM / recurrence launches of one step over the step time do not
        'hbm': hbm_roofline(),
        # `gflop` = ALGORITHMIC work (hidden size 200: the launches over the padded gate columns -- NP = 832 per direction for 4 H = 800 --
        # are counted at 800 / 832 of what they execute); `gflop_executed` = what the kernels multiply (round-5 verdict, item 4 iv)
        'step': (None if not (sampled_steps and ms_per_step) else (lambda tf: {'gflop': round(sum(v['flops'] for v in fam.values()) / sampled_steps / 1e9, 1),
                 'tflops': round(tf, 2), 'frac': round(tf / peak_tflops, 4),
                 'gflop_algorithmic': round(sum(v['flops'] for v in fam.values()) / sampled_steps / 1e9, 1),
                 'gflop_executed': round(sum(v.get('executed', v['flops']) for v in fam.values()) / sampled_steps / 1e9, 1)})(
                     sum(v['flops'] for v in fam.values()) / sampled_steps / (ms_per_step * 1e-3) / 1e12)),
    }


def _kernel_stats_file():
    """profiles/kernel_stats.json (tools/kernel_stats_json.py: the `rocprofv3 --kernel-trace --stats` tables of the headline command, in-step
    and with every stream collapsed into one, stamped with the build id) if it belongs to THIS build, else (None, why)."""
    import json
    from . import _lib
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'kernel_stats.json')
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None, 'profiles/kernel_stats.json missing or unreadable'
    have, now = d.get('build_id') or {}, _lib.build_id()
    if have.get('src_sha256') == now['src_sha256'] or (have.get('lib_sha256') and have.get('lib_sha256') == now['lib_sha256']):
        return d, 'build_id matches (%s)' % now['src_sha256']
    return None, 'profiles/kernel_stats.json was collected on another build (%s, running %s): not quoted' % (have.get('src_sha256'), now['src_sha256'])


def rocprof_block(roof, peak_tflops):
    """`roofline.rocprof`: the dominant kernel's average launch duration from the committed rocprofv3 tables of the same command -- inside
    the step (the launch shares the chip with up to four other HIP streams) and solo (NNR_ONE_STREAM=1) -- and the fractions of the fp32 MFMA
    peak they give for the ALGORITHMIC FLOPs per launch measured live, so that `frac` can be reproduced from profiles/ (round-5 verdict:
    0.178 in the line by HIP events that include queueing, 0.247 by rocprofv3's in-step average, 0.53 solo)."""
    d, why = _kernel_stats_file()
    out = {'source': 'profiles/kernel_stats.json <- rocprofv3 --kernel-trace --stats of `bench.py --steps 12 --warmup 4` (in_step) and the same under '
                     'NNR_ONE_STREAM=1 (solo); ' + why}
    if d is None or not roof or not roof.get('launches'):
        out.update(in_step_avg_us=None, solo_avg_us=None, frac_in_step=None, frac_solo=None)
        return out
    kern = roof['kernel']
    gflop_per_launch = roof['achieved'] * roof['avg_launch_us'] * 1e-3            # TFLOP/s x us = MFLOP -> x 1e-3 = GFLOP
    ins = (d.get('in_step') or {}).get(kern)
    solo = (d.get('solo') or {}).get(kern)
    out['kernel'] = kern
    out['gflop_per_launch_live'] = round(gflop_per_launch, 3)
    out['in_step_avg_us'] = None if not ins else ins['avg_us']
    out['solo_avg_us'] = None if not solo else solo['avg_us']
    out['frac_in_step'] = None if not ins else round(gflop_per_launch / (ins['avg_us'] * 1e-6) / 1e3 / peak_tflops, 4)
    out['frac_solo'] = None if not solo else round(gflop_per_launch / (solo['avg_us'] * 1e-6) / 1e3 / peak_tflops, 4)
    return out

