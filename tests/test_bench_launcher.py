"""`python bench.py --gpus N` must work as typed: with N > 1 and no torchrun environment it launches the N ranks itself (a
child `python -m torch.distributed.run`, before anything touches the GPU).  The CPU form of the run (`--plumbing_check`,
gloo) drives the PRODUCT's flat-buffer and exchange plumbing -- trainer.FlatParams, dp.init_from_env, dp.broadcast_parameters,
dp.GradientExchange (early + late bucket, 1/world scale) -- on a stand-in module; rank 0 prints one JSON line.
Reference semantics: trainer.py:212-220 (per-rank batch = batch_size // world_size, averaged gradients), :256-258."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, env=env, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    return r, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize('n', [1, 2])
def test_bench_self_launches_ranks_and_exchanges_gradients(n):
    r, out = _run(['--gpus', str(n), '--plumbing_check', '--steps', '3'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert out is not None and out['n_gpus'] == n and out['params_equal_on_all_ranks_and_expected'] is True
    # the headline follows the reference's flag semantics: --batch_size (default 64) is the GLOBAL batch, every rank takes
    # batch_size // world_size impressions (config.py:116, trainer.py:218) -- the same optimisation problem at every N
    assert out['scaling'] == 'strong' and out['global_batch'] == 64 and out['per_gpu_batch'] == 64 // n
    names = [b['name'] for b in out['buckets']['buckets']]
    assert names == ['early (user encoder)', 'late']


def test_bench_batch_flags_follow_the_reference_batch_semantics():
    """`--batch_size G` on N GPUs = the reference's `--batch_size G --world_size N`: per-rank batch G // N (trainer.py:218), an uneven
    remainder dropped as there; --weak makes G the per-GPU batch; --global_batch is the old spelling of --batch_size."""
    r, out = _run(['--gpus', '2', '--plumbing_check', '--steps', '1', '--batch_size', '48'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert out['scaling'] == 'strong' and out['per_gpu_batch'] == 24 and out['global_batch'] == 48
    r, out = _run(['--gpus', '2', '--plumbing_check', '--steps', '1', '--global_batch', '64'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert out['scaling'] == 'strong' and out['per_gpu_batch'] == 32 and out['global_batch'] == 64
    r, out = _run(['--gpus', '2', '--plumbing_check', '--steps', '1', '--weak'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert out['scaling'] == 'weak' and out['per_gpu_batch'] == 64 and out['global_batch'] == 128
    sys.path.insert(0, ROOT)
    import bench
    assert bench.shard_sizes(64, 8, False) == (8, 64) and bench.shard_sizes(64, 3, False) == (21, 63) and bench.shard_sizes(64, 8, True) == (64, 512)
    a = bench.parse(['--config', 'mhsa', '--gpus', '4'])
    assert (a.news_encoder, a.user_encoder, a.batch_size, a.weak) == ('MHSA', 'MHSA', 64, False)


def test_gradient_exchange_buckets_cover_the_flat_buffer_exactly_once():
    from nnr_amd import dp
    from nnr_amd.trainer import FlatParams, _Own
    from nnr_amd.config import make_config
    from oracle import nnr_oracle as O          # any module tree with the reference's names: news_encoder shared by user_encoder
    cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE'], corpus_sizes=dict(vocabulary_size=50), hidden_dim=16, attention_dim=8,
                      word_embedding_dim=12, category_embedding_dim=4, subCategory_embedding_dim=4, gcn_layer_num=2, category_num=5,
                      subCategory_num=7)
    model = O.Model(cfg)
    flat = FlatParams(model)
    ue = model.user_encoder
    own = [m for name, m in ue.named_children() if name != 'news_encoder']
    ex = dp.GradientExchange(flat, early_modules=[_Own(ue, own)])
    assert ex.early_span is not None
    a, b = ex.early_span
    own_ids = {id(p) for p in _Own(ue, own).parameters()}
    shared_ids = {id(p) for p in model.news_encoder.parameters()}
    assert own_ids and not (own_ids & shared_ids)
    covered = torch.zeros(flat.numel, dtype=torch.int32)
    covered[a:b] += 1
    for x, y in ex.late_spans:
        covered[x:y] += 1
    assert bool((covered == 1).all())
    for p, o in zip(flat.params, flat.offsets):          # the early span holds exactly the user encoder's own parameters
        assert (a <= o < b) == (id(p) in own_ids)
    # world size 1, not forced: nothing to exchange, scale 1
    flat.grad.fill_(2.0)
    ex.early_ready()
    assert ex.finish() == 1.0 and float(flat.grad.min()) == 2.0


def test_secondary_legs_name_every_other_baseline_config():
    """`bench.py --gpus 1` reports every other BASELINE.json config it can run on one GPU under `secondary` (round-4 verdict, item 3).
    CPU side of that contract: the leg table matches BASELINE.json's configs (encoder pair, dataset, global batch / GPU count ->
    per-GPU shard, trainer.py:218), the reference's per-dataset overrides apply (config.py:85-94), and the plumbing line says which
    keys a GPU run adds.  (The legs themselves need the MI355X; profiles/r05_bench.json is a driver-shaped run.)"""
    sys.path.insert(0, ROOT)
    import bench
    from nnr_amd.config import make_config
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))['configs']
    legs = {l[0]: l for l in bench.SECONDARY_LEGS}
    assert list(legs) == ['mhsa_mhsa_b64', 'cne_sue_shard_b8', 'cne_sue_large_shard_b16_v130000']
    # configs[1]: MHSA+MHSA, 200k, one GPU
    assert 'MHSA' in base[1] and legs['mhsa_mhsa_b64'][1:4] == ('MHSA', 'MHSA', '200k') and legs['mhsa_mhsa_b64'][4:7] == (64, 1, 64)
    # configs[3]: CNE+SUE, 200k, batch 64 over 8 GPUs -> 8 per GPU
    assert 'batch_size=64, 8' in base[3] and legs['cne_sue_shard_b8'][4:7] == (64, 8, 8)
    # configs[4]: CNE+SUE, large, batch 128 over 8 GPUs -> 16 per GPU, large vocabulary
    assert 'batch_size=128, 8' in base[4] and legs['cne_sue_large_shard_b16_v130000'][3:8] == ('large', 128, 8, 16, 130000)
    for name, ne, ue, dataset, gbatch, gworld, per_gpu, V in bench.SECONDARY_LEGS:
        cfg = make_config(['--news_encoder=' + ne, '--user_encoder=' + ue, '--dataset=' + dataset, '--batch_size=%d' % gbatch,
                           '--world_size=%d' % gworld], corpus_sizes=dict(vocabulary_size=V))
        assert cfg.batch_size // cfg.world_size == per_gpu and cfg.vocabulary_size == V
        assert abs(cfg.dropout_rate - (0.1 if dataset == 'large' else 0.2)) < 1e-12 and cfg.gcn_layer_num == 4
    a = bench.parse([])
    assert not a.no_secondary and a.secondary_steps == 10 and a.secondary_warmup >= 5      # >= 5: the timed steps are all native replays
    committed = os.path.join(ROOT, 'profiles', 'r06h_bench.json')
    if os.path.exists(committed):                      # the shape of a real GPU line (committed with the round's profiles)
        line = json.loads([l for l in open(committed) if l.startswith('{')][-1])
        sec = line['secondary']
        extra = {l[0]: l for l in bench.F32_ONLY_LEGS}
        assert set(legs) <= set(sec) <= set(legs) | set(extra)
        for name, leg in sec.items():
            assert 'error' not in leg, (name, leg)
            assert leg['ms_per_step'] > 0 and leg['value'] > 0 and 0 < leg['step']['frac'] < 1 and leg['per_gpu_batch'] == dict(legs, **extra)[name][6]
            assert leg['timed_window'].startswith('un-instrumented')
        # configs[4]'s leg prices its HBM-bound kernels at V = 130 000 (round-5 verdict, missing item 2)
        hbm = sec['cne_sue_large_shard_b16_v130000']['hbm']
        assert {'embed_gather', 'embed_scatter', 'sumsq', 'clip_adam'} <= set(hbm) and all(0 < v['frac'] <= 1 for v in hbm.values())
        m = sec['mhsa_mhsa_b64']['roofline_mhsa']
        assert m['mhsa_fwd']['mfma_tflops'] > 0 and m['mhsa_bwd']['mfma_tflops'] > 0
        assert line['cpu_baseline']['headline_batch']['batch'] == 64
        assert line['dtype'] == 'f32' and 'bf16x3' in line['config']['matrix_path']['nt_weight_gemms']
        r = line['roofline']
        assert set(r['rocprof']) >= {'in_step_avg_us', 'solo_avg_us', 'frac_in_step', 'frac_solo'} and r['step']['gflop_executed'] >= r['step']['gflop_algorithmic']
