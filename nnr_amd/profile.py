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
    'gemm_nt_pipe2_128x80': 'gemm_nt_pipe2_kernel<2, 5, 3, 2>',
    'gemm_tn_pipe2_128x80': 'gemm_tn_pipe2_kernel<2, 5, 3, 3>', 'gemm_tn_pipe2_128x208': 'gemm_tn_pipe2_kernel<2, 13, 3, 2>', 'gemm_tn_pipe2_128x160': 'gemm_tn_pipe2_kernel<2, 10, 3, 2>',
    'gemm_tn_pipe128x80': 'gemm_tn_pipe_kernel<2, 5, 3, 3, 0>', 'gemm_tn_pipe128x208': 'gemm_tn_pipe_kernel<2, 13, 3, 2, 0>',
    'gemm_nn_128x80': 'gemm_kernel<2, 5, false, true, 16>', 'gemm_tn_128x80': 'gemm_kernel<2, 5, true, true, 16>', 'lstm_fwd': 'lstm_fwd_pair_kernel<13>', 'lstm_bwd': 'lstm_bwd_pair_kernel<13>',
}


_enabled, _every = False, 1


def enable(every=1):
    """Start recording spans; with every > 1 only the steps announced by begin_step(i) with i % every == 0 are instrumented
    (two HIP events per launch cost ~5 % of a step when every launch of every step carries them)."""
    global _on, _enabled, _every
    _enabled, _every = True, max(1, int(every))
    _on = True
    _records.clear()


def begin_step(i):
    global _on
    _on = _enabled and (i % _every == 0)


def disable():
    global _on, _enabled
    _on = _enabled = False


def active():
    return _on


class span:
    """with profile.span(family, flops_fn): <launch>"""

    def __init__(self, family, flops_fn):
        self.family, self.flops_fn = family, flops_fn

    def __enter__(self):
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
    torch.cuda.synchronize()
    out = {}
    for family, s, e, fn in _records:
        tag = getattr(fn, 'tag', '')
        d = out.setdefault((family, tag), dict(ms=0.0, flops=0.0, launches=0))
        d['ms'] += s.elapsed_time(e)
        d['flops'] += float(fn())
        d['launches'] += 1
    return out


def summary():
    torch.cuda.synchronize()
    fam = {}
    for family, s, e, fn in _records:
        d = fam.setdefault(family, dict(ms=0.0, flops=0.0, launches=0))
        d['ms'] += s.elapsed_time(e)
        d['flops'] += float(fn())
        d['launches'] += 1
    return fam


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command (profiles/pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, corrected for gfx950 by tools/pmc_traffic.py), or None."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'pmc_traffic.json')
    try:
        for k in json.load(open(path))['kernels']:
            if k['kernel'] == kernel:
                return k['hbm_bytes_per_launch']
    except (OSError, ValueError, KeyError):
        pass
    return None


def tn_operand_bytes():
    """Per weight-gradient (token-reduction GEMM) family: algorithmic operand bytes per launch = live reduction rows x (M + N) x 4,
    over the recorded launches -- the figure the PMC traffic of profiles/pmc_traffic.json is compared with."""
    torch.cuda.synchronize()
    out = {}
    for family, s, e, fn in _records:
        dims = getattr(fn, 'tn_dims', None)
        if dims:
            M, N, fs = dims
            d = out.setdefault(family, dict(bytes=0.0, launches=0))
            d['bytes'] += float(fn()) / (2.0 * M * N * fs) * (M + N) * 4.0
            d['launches'] += 1
    return out


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
        'of this command on the same build (PMC cannot be collected inside the timed run), FETCH_SIZE x2 per the gfx950 note', 'kernel': KERNEL_OF.get(name, name), 'family': name, 'launches': d['launches'],
        'avg_launch_us': round(1000 * d['ms'] / max(1, d['launches']), 2),
        'share_of_instrumented_time': round(d['ms'] / total_ms, 3),
        'families': {k: {'ms': round(v['ms'], 3), 'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) if v['ms'] > 0 else 0.0,
                         'launches': v['launches']} for k, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms'])},
        # whole-step view: the launches of several HIP streams overlap, so per-kernel wall durations double-count the chip;
        # algorithmic FLOPs of all instrumented GEMM / recurrence launches of one step over the step time do not
        'step': (None if not (sampled_steps and ms_per_step) else (lambda tf: {'gflop': round(sum(v['flops'] for v in fam.values()) / sampled_steps / 1e9, 1),
                 'tflops': round(tf, 2), 'frac': round(tf / peak_tflops, 4)})(sum(v['flops'] for v in fam.values()) / sampled_steps / (ms_per_step * 1e-3) / 1e12)),
    }
