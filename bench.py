#!/usr/bin/env python3
"""Headline benchmark of the hot path: training impressions/sec, CNE+SUE on MIND-200k-shaped synthetic data, global
batch 64 (BASELINE.json), one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One "step" = one optimizer step of trainer.py:105-120 (forward, loss, backward, [RCCL all-reduce of the flat gradient],
clip_grad_norm_(4), Adam) on one batch that is already resident in HBM; dropout is ON (0.2, the reference's 200k
setting).

Scaling mode.  Impressions are independent units sharded over the ranks, so the default is WEAK scaling: every GPU
processes the headline batch of 64 impressions per step (global batch 64*N) and `value` = impressions of all ranks / time.
`--global_batch G` instead fixes the GLOBAL batch (the reference's `--batch_size` semantics: per-rank = G // world_size,
trainer.py:218) = strong scaling; at G=64 on 8 GPUs that is 8 impressions per GPU, a regime bound by the 128-step
dependent chain of the Bi-LSTM and by launch latency, not by throughput (measured on 1 GPU: 14.8 ms/step at batch 8 vs
24.8 ms at batch 64).

Prints ONE JSON line (rank 0) with the throughput, the roofline of the dominant kernel measured live with HIP events
on the launch stream, and a CPU baseline (the oracle, timed on this box's host cores on a bounded sample)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--news_encoder', default='CNE')
    ap.add_argument('--user_encoder', default='SUE')
    ap.add_argument('--batch_size', type=int, default=64, help='impressions per GPU per step (weak scaling, the default)')
    ap.add_argument('--global_batch', type=int, default=0, help='>0: strong scaling, this GLOBAL batch split over the GPUs')
    ap.add_argument('--vocabulary_size', type=int, default=60000)
    ap.add_argument('--dense', action='store_true', help='all titles/abstracts at full length (worst-case roofline variant)')
    ap.add_argument('--device_corpus', action='store_true', help='build every batch inside the timed step from the device-resident '
                    'corpus (id-only batches: nnr_corpus_batch + nnr_history_graph) instead of re-using pre-built batches')
    ap.add_argument('--roofline_every', type=int, default=4, help='instrument every n-th timed step with HIP events (the two events per '
                    'launch cost ~5 %% of a step when all steps carry them)')
    ap.add_argument('--zipf_s', type=float, default=None, help='diagnostic: exponent of the synthetic word-id distribution (default: SynthSpec)')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--cpu_baseline_batch', type=int, default=8)
    ap.add_argument('--cpu_baseline_steps', type=int, default=2)
    return ap.parse_args()


def cpu_baseline(cfg, spec, batch_size, steps):
    """Time the CPU oracle (oracle/nnr_oracle.py, pinned against the reference by tests/golden) on this box's host cores."""
    from nnr_amd.synth import SynthCorpus, to_torch
    from oracle import nnr_oracle as O
    # The oracle's explicit-time-loop LSTM is thousands of small ops: beyond ~16 threads intra-op parallelism stops helping
    # (measured on the 256-core bench host: 8 / 16 / 32 threads -> 4.7 / 4.3 / 4.7 s per batch-4 step; 256 threads stall).
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = O.Model(cfg)
    model.initialize()
    model.train()
    opt = O.make_optimizer(model, cfg)
    corpus = SynthCorpus(spec)
    rng = np.random.default_rng(11)
    O.train_step(model, opt, to_torch(corpus.batch(batch_size, rng)), cfg.gradient_clip_norm)          # warm-up
    # bounded sample: at least `steps` optimizer steps, more (up to 60) until ~10 s of CPU work are timed (a CNE+SUE step at
    # batch 8 takes ~10 s on 16 threads, an MHSA+MHSA step 0.2 s)
    done, dt = 0, 0.0
    while done < steps or (dt < 10.0 and done < 60):
        b = to_torch(corpus.batch(batch_size, rng))
        t0 = time.perf_counter()
        O.train_step(model, opt, b, cfg.gradient_clip_norm)
        dt += time.perf_counter() - t0
        done += 1
    return dict(value=round(done * batch_size / dt, 4), unit='impressions/s', cores=cores, kind='port',
                sample='%d optimizer steps of the same workload at batch %d (%.1f s of CPU work), torch %s CPU ops, %d threads' %
                       (done, batch_size, dt, torch.__version__, cores))


def main():
    a = parse()
    from nnr_amd import dp, ops
    from nnr_amd.config import make_config
    from nnr_amd.model import Model
    from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
    from nnr_amd.trainer import Trainer
    from nnr_amd import profile as prof

    rank, local, world = dp.init_from_env('nccl' if a.gpus > 1 else None)
    assert world == a.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % a.gpus
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    weak = a.global_batch <= 0
    per_gpu = a.batch_size if weak else a.global_batch // world
    global_batch = per_gpu * world
    cfg = make_config(['--news_encoder=' + a.news_encoder, '--user_encoder=' + a.user_encoder, '--dataset=200k',
                       '--batch_size=%d' % global_batch, '--world_size=%d' % world],
                      corpus_sizes=dict(vocabulary_size=a.vocabulary_size))
    spec = SynthSpec(vocabulary_size=cfg.vocabulary_size, dense=a.dense, **({} if a.zipf_s is None else dict(zipf_s=a.zipf_s)))

    torch.manual_seed(cfg.seed)
    table = torch.randn(cfg.vocabulary_size, cfg.word_embedding_dim) * 0.3
    table[0] = 0
    model = Model(cfg, table)
    model.initialize()
    model = model.to(dev).train()
    trainer = Trainer(model, cfg)

    corpus = SynthCorpus(spec)
    rng = np.random.default_rng(100 + rank)
    nb = min(8, a.steps + a.warmup)
    if a.device_corpus:
        from nnr_amd.corpus import from_synth
        dcorpus = from_synth(corpus, 4096, rng, dev, graph='build')
        order = [torch.from_numpy(rng.permutation(4096)[:per_gpu].astype(np.int32)).to(dev) for _ in range(nb)]

        def fresh(i):      # 256 bytes of behaviour ids per batch; the 21 tensors are gathered / built in HBM
            return dcorpus.train_batch(order[i % nb])
    else:
        batches = [to_torch(corpus.batch(per_gpu, rng), dev) for _ in range(nb)]

        def fresh(i):      # masks are mutated in place by the model; mutation is idempotent, so batches can be reused
            return batches[i % nb]

    for i in range(a.warmup):
        trainer.train_step(fresh(i))
    dp.barrier()
    torch.cuda.synchronize()
    prof.enable(every=a.roofline_every)     # live HIP-event spans on every `roofline_every`-th step of the timed region
    t0 = time.perf_counter()
    for i in range(a.steps):
        prof.begin_step(i)
        trainer.train_step(fresh(a.warmup + i))
    dp.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof.disable()
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax)
    sampled = len(range(0, a.steps, max(1, a.roofline_every)))
    roof = prof.roofline(PEAK_F32_TFLOPS, sampled_steps=sampled, ms_per_step=1000 * dt / a.steps)
    if roof and (a.news_encoder, a.user_encoder, per_gpu, a.dense) != ('CNE', 'SUE', 64, False):
        roof['traffic'] = None            # the committed PMC passes (profiles/pmc_traffic.json) are of the headline command only
    exchange_timeouts = ops.lstm_sync_timeouts()      # the CU-pair recurrence's exchange must never time out (last launch's counter)
    if exchange_timeouts:
        print('WARNING: pair-recurrence exchange timed out %d times (values poisoned with NaN)' % exchange_timeouts, file=sys.stderr)

    if rank == 0:
        out = {
            'metric': 'training impressions/sec on MIND-200k (CNE+SUE, bs=64)' if (a.news_encoder, a.user_encoder) == ('CNE', 'SUE')
                      else 'training impressions/sec (%s+%s)' % (a.news_encoder, a.user_encoder),
            'value': round(a.steps * global_batch / dt, 2), 'unit': 'impressions/s', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': round(1000 * dt / a.steps, 3), 'higher_is_better': True,
            'scaling': 'weak' if weak else 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s+%s train step, MIND-200k-shaped synthetic batches, dropout %.2f, gcn_layer_num %d%s' %
                                   (a.news_encoder, a.user_encoder, cfg.dropout_rate, cfg.gcn_layer_num, ', dense lengths' if a.dense else ''),
                       'global_batch': global_batch, 'per_gpu_batch': per_gpu, 'parallelism': 'dp%d' % world,
                       'synth': {k: v for k, v in spec.describe().items() if k in ('vocabulary_size', 'title_len_mean', 'content_len_mean', 'news_pool', 'dense')},
                       'batches': 'device-resident corpus, id-only' if a.device_corpus else 'pre-built, resident in HBM',
                       'peak_hbm_reserved_gb': round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 2),
                       'recurrence_exchange_timeouts': exchange_timeouts},
            'roofline': roof,
        }
        if not a.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(cfg, spec, a.cpu_baseline_batch, a.cpu_baseline_steps)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
