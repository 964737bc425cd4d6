"""Arithmetic of bench.py's `roofline` object on the CPU (nnr_amd/profile.py): the records are what a timing replay of the launch
tape delivers -- (kernel family, algorithmic-FLOPs closure, milliseconds) -- so the assembly of the line can be pinned without
a GPU: dominant family, achieved = FLOPs / time, per-launch averages, algorithmic operand bytes, the whole-step view, and the gate
that keeps counter traffic of ANOTHER build out of the line (profiles/pmc_traffic.json carries the build id it was collected on)."""
import json
import os

import pytest
import torch

from nnr_amd import _lib, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 157.3


def _fn(flops, op_bytes=None, tn_dims=None):
    def f(vals=None):
        return flops
    if op_bytes is not None:
        f.op_bytes = op_bytes
    if tn_dims is not None:
        f.tn_dims = tn_dims
    return f


@pytest.fixture
def records(monkeypatch):
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
    profile.enable(every=1, eager=False)
    M, N, K = 1664, 200, 22000                                   # a weight-gradient launch: K = live token rows
    tn = _fn(2.0 * M * N * K, op_bytes=(K * (M + N) + M * N) * 4.0, tn_dims=(M, N, 1))
    nt = _fn(9.0e9, op_bytes=5.0e7)
    profile.TAPE_RECORDS.extend([('gemm_tn_pipe2_64x208', tn, 0.5), ('gemm_nt_pipe2_128x80', nt, 0.3), ('gemm_tn_pipe2_64x208', tn, 0.7),
                                 ('lstm_bwd', _fn(3.4e10), 0.2)])
    profile._PMC.clear()
    yield dict(M=M, N=N, K=K, tn_flops=2.0 * M * N * K)
    profile.disable()
    profile.TAPE_RECORDS.clear()
    profile._PMC.clear()


def test_roofline_arithmetic_and_dominant_family(records):
    r = profile.roofline(PEAK, sampled_steps=1, ms_per_step=2.0)
    f = records['tn_flops']
    assert r['family'] == 'gemm_tn_pipe2_64x208' and r['kernel'] == 'gemm_tn_pipe2_kernel<1, 13, 3, 2>'      # most instrumented time
    assert r['launches'] == 2 and r['avg_launch_us'] == 600.0 and r['bound'] == 'mfma' and r['peak'] == PEAK
    ach = 2 * f / 1.2e-3 / 1e12
    assert r['achieved'] == round(ach, 3) and r['frac'] == round(ach / PEAK, 4)
    assert r['share_of_instrumented_time'] == round(1.2 / 1.7, 3)
    assert list(r['families']) == ['gemm_tn_pipe2_64x208', 'gemm_nt_pipe2_128x80', 'lstm_bwd']                # by time, descending
    assert r['families']['gemm_nt_pipe2_128x80'] == {'ms': 0.3, 'tflops': 30.0, 'launches': 1}
    total = 2 * f + 9.0e9 + 3.4e10
    assert r['step'] == {'gflop': round(total / 1e9, 1), 'tflops': round(total / 2.0e-3 / 1e12, 2), 'frac': round(total / 2.0e-3 / 1e12 / PEAK, 4),
                         'gflop_algorithmic': round(total / 1e9, 1), 'gflop_executed': round(total / 1e9, 1)}
    assert r['algorithmic_bytes_per_launch'] == round((records['K'] * (records['M'] + records['N']) + records['M'] * records['N']) * 4.0)
    # operand bytes of the token reduction alone (what the counter traffic of a weight-gradient family is compared with)
    tn = profile.tn_operand_bytes()['gemm_tn_pipe2_64x208']
    assert tn['launches'] == 2 and tn['bytes'] == pytest.approx(2 * records['K'] * (records['M'] + records['N']) * 4.0)


def test_executed_flops_beside_the_algorithmic_ones(records):
    """A launch over padded gate columns (NP = 832 for 4 H = 800) carries scale = 800 / 832: `gflop` / `gflop_algorithmic` count the hidden
    size the model has, `gflop_executed` what the kernel multiplies (round-5 verdict, item 4 iv)."""
    f = _fn(8.0e9)
    f.scale = 800.0 / 832.0
    profile.TAPE_RECORDS.append(('gemm_nt_bx3_128x80', f, 0.1))
    r = profile.roofline(PEAK, sampled_steps=1, ms_per_step=2.0)
    total = 2 * records['tn_flops'] + 9.0e9 + 3.4e10 + 8.0e9
    assert r['step']['gflop_algorithmic'] == r['step']['gflop'] == round(total / 1e9, 1)
    assert r['step']['gflop_executed'] == round((total + 8.0e9 * (832.0 / 800.0 - 1.0)) / 1e9, 1)
    assert profile.KERNEL_OF['gemm_nt_bx3_128x80'] == 'gemm_nt_bx3_kernel<2, 5, 2>'


def test_rocprof_block_reproduces_frac_from_the_committed_tables(records, monkeypatch, tmp_path):
    """`roofline.rocprof` = {in_step_avg_us, solo_avg_us, frac_in_step, frac_solo}: the dominant kernel's rocprofv3 averages from
    profiles/kernel_stats.json (build-id gated, like the counter traffic) against the algorithmic FLOPs per launch measured live."""
    r = profile.roofline(PEAK, sampled_steps=1, ms_per_step=2.0)
    d = json.load(open(os.path.join(ROOT, 'profiles', 'kernel_stats.json')))
    assert set(d['build_id']) == {'src_sha256', 'lib_sha256'} and len(d['in_step']) >= 20 and len(d['solo']) >= 20
    kern = r['kernel']
    assert kern in d['in_step'] and kern in d['solo']
    monkeypatch.setattr(_lib, 'build_id', lambda: {'src_sha256': 'x' * 16, 'lib_sha256': 'y' * 16})
    b = profile.rocprof_block(r, PEAK)
    assert b['in_step_avg_us'] is None and b['frac_solo'] is None and 'collected on another build' in b['source']
    monkeypatch.setattr(_lib, 'build_id', lambda: d['build_id'])
    b = profile.rocprof_block(r, PEAK)
    g = records['tn_flops'] / 1e9                                  # GFLOP per launch of the dominant family
    assert b['kernel'] == kern and b['gflop_per_launch_live'] == pytest.approx(g, rel=1e-3)
    assert b['in_step_avg_us'] == d['in_step'][kern]['avg_us'] and b['solo_avg_us'] == d['solo'][kern]['avg_us']
    assert b['frac_in_step'] == pytest.approx(g / (b['in_step_avg_us'] * 1e-6) / 1e3 / PEAK, rel=2e-3)
    assert b['frac_solo'] == pytest.approx(g / (b['solo_avg_us'] * 1e-6) / 1e3 / PEAK, rel=2e-3)
    assert b['frac_solo'] > b['frac_in_step'] > 0


def test_bench_line_carries_the_round6_keys():
    """Shape of what bench.py adds in round 6 (no GPU: the helpers only): config.matrix_path names the split and the pure-f32 leg; the
    N > 1 line quotes the batch-sweep ceiling; secondary legs exist for every other BASELINE config + the pure fp32-MFMA leg."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from nnr_amd import ops
    mp = bench.matrix_path(ops, {(450560, 400, 400, 'weight'): 2})
    if ops.BX3[0]:
        assert 'bf16x3' in mp['nt_weight_gemms'] and mp['pure_f32_leg'] == 'secondary.f32_mfma_only_cne_sue_b64' and mp['launch_classes'] == {'450560x400x400 weight': 2}
    else:
        assert mp['nt_weight_gemms'] == 'v_mfma_f32_16x16x4_f32'
    assert mp['weight_gradient_gemms'] == 'v_mfma_f32_16x16x4_f32'
    assert [l[0] for l in bench.SECONDARY_LEGS] == ['mhsa_mhsa_b64', 'cne_sue_shard_b8', 'cne_sue_large_shard_b16_v130000']
    assert [l[0] for l in bench.F32_ONLY_LEGS] == ['f32_mfma_only_cne_sue_b64']
    c = bench.scaling_ceiling(8, 64)
    sweep = json.load(open(os.path.join(ROOT, 'profiles', 'batch_sweep.json')))
    assert c['per_gpu_batch'] == 8 and c['one_gpu_ms_per_step_at_that_batch'] == sweep['ms_per_step']['8']
    assert c['value_ceiling'] == round(64 / sweep['ms_per_step']['8'] * 1000.0, 1)
    a = bench.parse(['--roofline_every', '7'])
    assert a.roofline_steps == 2                                   # the timed window carries no events; two extra steps after it do


def test_counter_traffic_is_quoted_only_for_the_build_it_was_collected_on(records, monkeypatch):
    d = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
    have = d['build_id']
    per_launch = {k['kernel']: k['hbm_bytes_per_launch'] for k in d['kernels']}
    want = per_launch['gemm_tn_pipe2_kernel<1, 13, 3, 2>']
    # another build: neither the sources' nor the binary's hash matches
    monkeypatch.setattr(_lib, 'build_id', lambda: {'src_sha256': 'x' * 16, 'lib_sha256': 'y' * 16})
    r = profile.roofline(PEAK, sampled_steps=1, ms_per_step=2.0)
    assert r['traffic'] is None and r['traffic_over_algorithmic'] is None and 'collected on another build' in r['traffic_source']
    assert profile.weight_gradient_traffic() is None
    # this build by its sources (the binary may have been re-linked), and by its binary alone
    for now in ({'src_sha256': have['src_sha256'], 'lib_sha256': 'y' * 16}, {'src_sha256': 'x' * 16, 'lib_sha256': have['lib_sha256']}):
        profile._PMC.clear()
        monkeypatch.setattr(_lib, 'build_id', lambda now=now: now)
        r = profile.roofline(PEAK, sampled_steps=1, ms_per_step=2.0)
        assert r['traffic'] == want and 'build_id matches' in r['traffic_source']
        assert r['traffic_over_algorithmic'] == round(want / r['algorithmic_bytes_per_launch'], 2)
        w = profile.weight_gradient_traffic()
        op = records['K'] * (records['M'] + records['N']) * 4.0
        assert w['gemm_tn_pipe2_64x208'] == {'operand_mb_per_launch': round(op / 1e6, 1), 'counter_mb_per_launch': round(want / 1e6, 1),
                                              'ratio': round(want / op, 2)}
        assert w['all'] == {'ratio': round(want / op, 2)}


def test_committed_counter_file_is_well_formed():
    """profiles/pmc_traffic.json: a build id and, per kernel, fetch (already x2 for gfx950's 32-byte unit) + write = total bytes."""
    d = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
    assert set(d['build_id']) == {'src_sha256', 'lib_sha256'} and len(d['kernels']) >= 20
    for k in d['kernels']:
        assert k['launches'] > 0 and k['hbm_bytes_per_launch'] == pytest.approx(k['fetch_bytes_per_launch'] + k['write_bytes_per_launch'], abs=2)
    names = {k['kernel'] for k in d['kernels']}
    assert {'gemm_tn_pipe2_kernel<1, 13, 3, 2>', 'gemm_nt_bx3_kernel<2, 5, 2>', 'lstm_bwd_pair_kernel<13, true>'} <= names
    # every kernel-name prefix `roofline.hbm` prices traffic by exists in the committed pass of the headline step (round 6: the pools' packed
    # kernel was renamed and the table kept pricing `pool_bwd` by the user encoder's small dense pool alone: 0.46 x its algorithmic bytes)
    for family, (prefixes, _) in profile.HBM_KERNELS.items():
        if family in ('gate_bwd', 'embed_scatter'):      # (fused into the GEMM epilogue / one of two alternative kernels)
            assert any(any(n.startswith(p) for n in names) for p in prefixes) or family == 'gate_bwd'
            continue
        for p in prefixes:
            assert any(n.startswith(p) for n in names), (family, p)


def test_hbm_families_get_their_own_roofline_and_stay_out_of_the_mfma_one(records, monkeypatch):
    """`roofline.hbm`: gathers / scatters / pools / element-wise / optimizer launches carry ALGORITHMIC bytes instead of FLOPs; they are
    priced against the 8 TB/s HBM peak, listed by time, and never become the `dominant` (MFMA) family or add to the step's FLOPs."""
    def hb(nbytes):
        f = _fn(0.0, op_bytes=nbytes)
        f.hbm = True
        return f
    profile.TAPE_RECORDS.extend([('embed_gather', hb(2.0e8), 0.05), ('embed_gather', hb(0.5e8), 0.02), ('clip_adam', hb(7.17e8), 5.0),
                                 ('sue_intra_bwd', hb(1.0e8), 0.1)])
    monkeypatch.setattr(_lib, 'build_id', lambda: {'src_sha256': 'x' * 16, 'lib_sha256': 'y' * 16})       # a build the committed counters are not of
    r = profile.roofline(PEAK, sampled_steps=1, ms_per_step=2.0)
    assert r['family'] == 'gemm_tn_pipe2_64x208' and list(r['families']) == ['gemm_tn_pipe2_64x208', 'gemm_nt_pipe2_128x80', 'lstm_bwd']
    assert r['share_of_instrumented_time'] == round(1.2 / 1.7, 3)
    h = r['hbm']
    assert list(h) == ['clip_adam', 'sue_intra_bwd', 'embed_gather']
    g = h['embed_gather']
    assert g['launches'] == 2 and g['avg_launch_us'] == 35.0 and g['algorithmic_bytes_per_launch'] == round(1.25e8)
    assert g['achieved'] == round(2.5e8 / 0.07e-3 / 1e9, 1) and g['peak'] == 8000.0 and g['unit'] == 'GB/s'
    assert g['frac'] == round(2.5e8 / 0.07e-3 / 1e9 / 8000.0, 4) and g['traffic'] is None          # (no PMC file for this fake build)
    # counter bytes per CALL from the committed file when the build matches: one call of sue_intra_bwd launches two kernels
    d = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
    profile._PMC.clear()
    monkeypatch.setattr(_lib, 'build_id', lambda: d['build_id'])
    per = {k['kernel']: (k['hbm_bytes_per_launch'], k['launches']) for k in d['kernels']}
    names = [n for n in per if n.startswith(('sue_intra_bwd_ds_kernel', 'sue_intra_bwd_dg_kernel'))]
    if len(names) == 2:
        want = sum(per[n][0] * per[n][1] for n in names) / (sum(per[n][1] for n in names) / 2)
        assert profile.pmc_traffic_family('sue_intra_bwd') == round(want)
    assert profile.pmc_traffic_family('no_such_family') is None
