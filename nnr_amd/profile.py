"""Live per-kernel-family timing for bench.py: HIP events recorded on the launch stream around each instrumented
launch (the kernels run on torch's current stream, so torch.cuda.Event sees them), plus the ALGORITHMIC FLOPs of the
launch (true dims, not the padded tile dims; data-dependent token counts are read back from device memory after the
timed region).  Families map 1:1 to kernel symbols, so the rocprofv3 --kernel-trace --stats averages under profiles/
can be compared directly."""
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
    'gemm_nt_pipe2_128x80': 'gemm_nt_pipe2_kernel<2, 5, 3, 2>', 'gemm_nt_pipe2_128x64': 'gemm_nt_pipe2_kernel<2, 4, 3, 2>',
    'gemm_tn_pipe2_128x80': 'gemm_tn_pipe2_kernel<2, 5, 3, 3>', 'gemm_tn_pipe2_128x208': 'gemm_tn_pipe2_kernel<2, 13, 3, 2>', 'gemm_tn_pipe2_128x160': 'gemm_tn_pipe2_kernel<2, 10, 3, 2>',
    'gemm_tn_pipe128x80': 'gemm_tn_pipe_kernel<2, 5, 3, 3, 0>', 'gemm_tn_pipe128x208': 'gemm_tn_pipe_kernel<2, 13, 3, 2, 0>',
    'gemm_nn_128x80': 'gemm_kernel<2, 5, false, true, 16>', 'gemm_tn_128x80': 'gemm_kernel<2, 5, true, true, 16>', 'lstm_fwd': 'lstm_fwd_pair_kernel<13>', 'lstm_bwd': 'lstm_bwd_pair_kernel<13>',
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


def summary():
    fam = {}
    for family, fn, ms in _all_records():
        d = fam.setdefault(family, dict(ms=0.0, flops=0.0, launches=0))
        d['ms'] += ms
        d['flops'] += float(fn())
        d['launches'] += 1
    return fam


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
        'step': (None if not (sampled_steps and ms_per_step) else (lambda tf: {'gflop': round(sum(v['flops'] for v in fam.values()) / sampled_steps / 1e9, 1),
                 'tflops': round(tf, 2), 'frac': round(tf / peak_tflops, 4)})(sum(v['flops'] for v in fam.values()) / sampled_steps / (ms_per_step * 1e-3) / 1e12)),
    }
