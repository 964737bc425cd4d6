#!/usr/bin/env python3
"""Headline benchmark of the hot path: training impressions/sec, CNE+SUE on MIND-200k-shaped synthetic data, global batch 64
(BASELINE.json), one process per GPU.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process is only a LAUNCHER: before anything touches the GPU it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` as a child
process and exits with the child's return code (a fresh child, never an exec of a GPU-touched process).  Started by torchrun
(RANK / LOCAL_RANK / WORLD_SIZE set) it is one rank of the job.

One "step" = one optimizer step of trainer.py:83-120 on one batch: batch delivery (the 21 tensors of trainer.py:83-103 are
gathered / built in HBM from the device-resident corpus out of 256 bytes of behaviour ids -- nnr_corpus_batch +
nnr_history_graph, a FRESH batch every step), forward, loss, backward, [RCCL all-reduce of the flat gradient],
clip_grad_norm_(4), Adam; dropout is ON (0.2, the reference's 200k setting).  `--prebuilt` re-uses eight pre-built batches
resident in HBM instead (the timed region of rounds 1-3).

Scaling.  `--batch_size 64` is the GLOBAL batch, exactly as the reference's flag (config.py:116): with N GPUs every rank
processes batch_size // N impressions per step (trainer.py:218) and the gradients are averaged over the ranks -- the SAME
optimisation problem at every N, so the headline line is STRONG scaling at global batch 64 (`scaling: "strong"`; BASELINE.json
configs[3] = per-GPU batch 8 at N = 8: a regime bound by the 128-step dependent chain of the Bi-LSTM and by launch latency,
not by throughput).  For N > 1 the same run also times WEAK scaling (64 impressions per GPU, global batch 64 N) and reports it
in the secondary `weak_scaling` object; `--weak` makes that the headline instead (`--no_weak` skips the leg).

Prints ONE JSON line (rank 0): throughput, the roofline of the dominant kernel measured live with HIP events on the launch
stream (+ `roofline.hbm`: the HBM-bound kernels of the step against the 8 TB/s peak), and a CPU baseline (the oracle with
ATen's packed-sequence LSTM = the reference's nn.LSTM host path, timed on this box's cores on a bounded sample).
`--config mhsa` runs BASELINE.json configs[1] (MHSA+MHSA) and adds the attention kernels' MFMA figures (`roofline.mhsa`).
Exit code 3 if the CU-pair recurrence's exchange ever timed out (values poisoned)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--config', choices=['cne_sue', 'mhsa'], default=None, help='shorthand for the encoder pair: cne_sue = BASELINE.json '
                    'configs[2..4] (the headline), mhsa = configs[1] (MHSA+MHSA, MFMA attention kernel)')
    ap.add_argument('--news_encoder', default='CNE')
    ap.add_argument('--user_encoder', default='SUE')
    ap.add_argument('--batch_size', type=int, default=64, help='GLOBAL batch (the reference\'s flag, config.py:116); every rank processes '
                    'batch_size // world_size impressions per step (trainer.py:218)')
    ap.add_argument('--weak', action='store_true', help='headline = weak scaling: --batch_size impressions PER GPU (global batch_size x N)')
    ap.add_argument('--no_weak', action='store_true', help='N > 1: skip the secondary weak-scaling (batch_size per GPU) leg')
    ap.add_argument('--global_batch', type=int, default=0, help='deprecated alias of --batch_size (rounds 2-3)')
    ap.add_argument('--prebuilt', action='store_true', help='re-use eight pre-built batches resident in HBM (rounds 1-3) instead of building a '
                    'fresh batch from the device-resident corpus inside every timed step')
    ap.add_argument('--vocabulary_size', type=int, default=60000)
    ap.add_argument('--dense', action='store_true', help='all titles/abstracts at full length (worst-case roofline variant)')
    ap.add_argument('--device_corpus', action='store_true', help='(default since round 4; kept for old command lines) build every batch '
                    'inside the timed step from the device-resident corpus (id-only batches: nnr_corpus_batch + nnr_history_graph)')
    ap.add_argument('--roofline_steps', type=int, default=2, help='instrumented steps (HIP events around every GEMM / recurrence / HBM-bound call) run AFTER the '
                    'timed window: the window itself and the secondary legs time un-instrumented replays (round-5 verdict: per-call events cost ~1.2 ms per '
                    'instrumented batch-64 step and 15 %% of a batch-8 leg)')
    ap.add_argument('--roofline_every', type=int, default=0, help='(rounds 1-5: instrument every n-th step INSIDE the timed window) accepted and ignored')
    ap.add_argument('--zipf_s', type=float, default=None, help='diagnostic: exponent of the synthetic word-id distribution (default: SynthSpec)')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_isolated', action='store_true', help='skip the two serialised extra steps behind `roofline.isolated`')
    ap.add_argument('--sustained_seconds', type=float, default=3.0, help='after the K timed steps, keep stepping for this long (same '
                    'workload, no instrumentation) and report it as the `sustained` object: clocks under a multi-second load; 0 = skip')
    ap.add_argument('--no_secondary', action='store_true', help='N = 1: skip the short secondary legs over the other BASELINE.json configs '
                    '(`secondary`: mhsa_mhsa_b64, cne_sue_shard_b8, cne_sue_large_shard_b16_v130000)')
    ap.add_argument('--no_experimental', '--no_f32_leg', dest='no_experimental', action='store_true', help='skip the pure fp32-MFMA leg of `secondary`')
    ap.add_argument('--secondary_steps', type=int, default=10)
    ap.add_argument('--secondary_warmup', type=int, default=5, help='two call-by-call steps + the recording + one replay + the discarded timing '
                    'replay: the timed steps of a secondary leg are all native replays, like the headline window')
    ap.add_argument('--cpu_baseline_headline_threads', default='32', help='comma-separated intra-op thread counts probed at the headline batch')
    ap.add_argument('--cpu_baseline_batch', type=int, default=8)
    ap.add_argument('--cpu_baseline_headline_steps', type=int, default=1, help='extra CPU-baseline steps at the HEADLINE batch (own thread probe), '
                    'reported beside the bounded batch-8 sample; 0 = skip')
    ap.add_argument('--cpu_baseline_steps', type=int, default=2)
    ap.add_argument('--plumbing_check', action='store_true', help='CPU only (gloo): run the launcher + the product\'s flat-buffer / '
                    'exchange plumbing (trainer.FlatParams, nnr_amd.dp) on a stand-in module and print one JSON line; no HIP call')
    a = ap.parse_args(argv)
    if a.config == 'mhsa':
        a.news_encoder = a.user_encoder = 'MHSA'
    elif a.config == 'cne_sue':
        a.news_encoder, a.user_encoder = 'CNE', 'SUE'
    if a.global_batch > 0:
        a.batch_size, a.weak = a.global_batch, False
    return a


def shard_sizes(batch_size, world, weak):
    """(per-GPU batch, global batch) of a run: the reference's semantics (`--batch_size` global, per rank batch_size // world,
    trainer.py:218; the remainder of an uneven division is dropped exactly as there) or, with --weak, batch_size per GPU."""
    per = batch_size if weak else batch_size // world
    return per, per * world


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a):
    """N > 1 and not yet inside a torchrun job: start the N ranks as a child process group and return its exit code.
    Nothing in this process has touched the GPU (no HIP call, no torch.cuda.is_available())."""
    # Under a profiler that preloads its library into this process (rocprofv3: with --pmc the GPU is initialised before main() runs)
    # the statement above is false, and starting the ranks from here would be an exec hop from a GPU-initialised process, which
    # this pool forbids (it takes the machine down).  Multi-GPU runs are not to be profiled through the launcher: profile ONE rank
    # (`rocprofv3 ... -- python3 bench.py --gpus 1`), or start the ranks with torch.distributed.run yourself.
    if any('rocprof' in v.lower() for v in (os.environ.get('LD_PRELOAD', ''), os.environ.get('ROCP_TOOL_LIBRARIES', ''), os.environ.get('HSA_TOOLS_LIB', ''))) \
            or any(k.startswith(('ROCPROF', 'ROCPROFILER_')) for k in os.environ):
        print('bench.py --gpus %d: refusing to launch the ranks from a process a profiler has attached to (see launch_ranks)' % a.gpus, file=sys.stderr)
        return 5
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL across processes needs it on this driver
    return subprocess.call(cmd, env=env)


def cpu_baseline(cfg, spec, batch_size, steps, headline_batch=0, headline_steps=1, headline_threads='32'):
    """Time the CPU oracle (oracle/nnr_oracle.py, pinned against the reference by tests/golden) on this box's host cores.
    The Bi-LSTM runs through ATen's own packed-sequence LSTM -- the code path the reference's nn.LSTM takes on the host
    (newsEncoders.py:119-127) -- not through the oracle's explicit time loop (2.8x slower, kept as the parity checker)."""
    import numpy as np
    import torch
    from nnr_amd.synth import SynthCorpus, to_torch
    from oracle import nnr_oracle as O
    O.BiLSTM.backend = 'aten'
    torch.manual_seed(0)
    model = O.Model(cfg)
    model.initialize()
    model.train()
    opt = O.make_optimizer(model, cfg)
    corpus = SynthCorpus(spec)
    rng = np.random.default_rng(11)
    # thread count: the step is a chain of mid-size CPU GEMMs; probe a few intra-op thread counts with one step each (the
    # first also warms up) and time the sample with the fastest
    ncpu = os.cpu_count() or 1
    cands = sorted({min(ncpu, c) for c in (8, 16, 32, 64)})
    best, probe = None, {}
    for c in cands:
        torch.set_num_threads(c)
        if best is None:
            O.train_step(model, opt, to_torch(corpus.batch(batch_size, rng)), cfg.gradient_clip_norm)      # warm-up
        t0 = time.perf_counter()
        O.train_step(model, opt, to_torch(corpus.batch(batch_size, rng)), cfg.gradient_clip_norm)
        probe[c] = time.perf_counter() - t0
        if best is None or probe[c] < probe[best]:
            best = c
    torch.set_num_threads(best)
    # bounded sample: at least `steps` optimizer steps, more (up to 60) until ~10 s of CPU work are timed
    done, dt = 0, 0.0
    while done < steps or (dt < 10.0 and done < 60):
        b = to_torch(corpus.batch(batch_size, rng))
        t0 = time.perf_counter()
        O.train_step(model, opt, b, cfg.gradient_clip_norm)
        dt += time.perf_counter() - t0
        done += 1
    res = dict(value=round(done * batch_size / dt, 4), unit='impressions/s', cores=best, kind='port',
               sample='%d optimizer steps of the same workload at batch %d (%.1f s of CPU work), oracle with ATen packed-sequence LSTM '
                      '(= the reference\'s nn.LSTM host path), torch %s, %d of %d cores (probe s/step: %s)' %
                      (done, batch_size, dt, torch.__version__, best, ncpu, {k: round(v, 2) for k, v in probe.items()}))
    if headline_batch and headline_batch != batch_size and headline_steps > 0:
        # the same oracle at the HEADLINE batch with its own thread probe (round-4 verdict: do more host cores help the reference
        # there?): one step per candidate thread count (--cpu_baseline_headline_threads, default 32 only: a batch-64 oracle step is 29 s on 32 and 41 s on 8 threads of the 256-core host, profiles/r05a_bench.json; the batch-8 sample above warmed the process up); the best probe step counts as the first of `headline_steps` timed steps
        hb, hprobe = None, {}
        for c in sorted({min(ncpu, int(c)) for c in str(headline_threads).split(',')}):
            torch.set_num_threads(c)
            t0 = time.perf_counter()
            O.train_step(model, opt, to_torch(corpus.batch(headline_batch, rng)), cfg.gradient_clip_norm)
            hprobe[c] = time.perf_counter() - t0
            if hb is None or hprobe[c] < hprobe[hb]:
                hb = c
        torch.set_num_threads(hb)
        hdt = hprobe[hb]                          # (a batch-64 oracle step is ~20 s: the best probe step is the first timed step)
        for _ in range(headline_steps - 1):
            b = to_torch(corpus.batch(headline_batch, rng))
            t0 = time.perf_counter()
            O.train_step(model, opt, b, cfg.gradient_clip_norm)
            hdt += time.perf_counter() - t0
        res['headline_batch'] = dict(batch=headline_batch, value=round(headline_steps * headline_batch / hdt, 4), unit='impressions/s', cores=hb,
                                     steps=headline_steps, seconds=round(hdt, 2), probe_s_per_step={k: round(v, 2) for k, v in hprobe.items()})
        res['sample'] += '; + %d step(s) at the headline batch %d on %d cores: %.2f impressions/s (probe s/step: %s)' % (
            headline_steps, headline_batch, hb, res['headline_batch']['value'], res['headline_batch']['probe_s_per_step'])
    return res


def plumbing_check(a):
    """CPU stand-in run of everything around the HIP model: launcher -> torchrun env -> dp.init_from_env(gloo) ->
    trainer.FlatParams -> dp.broadcast_parameters -> per-rank gradients -> dp.GradientExchange (bucketed all-reduce, 1/world)
    -> a plain SGD update on the flat buffer; rank 0 prints one JSON line a test can check."""
    import torch
    from nnr_amd import dp
    from nnr_amd.trainer import FlatParams
    rank, local, world = dp.init_from_env('gloo')
    assert world == a.gpus, 'world size %d != --gpus %d' % (world, a.gpus)
    torch.manual_seed(1234 + rank)                               # different init per rank: the broadcast must fix it
    head = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    tail = torch.nn.Linear(3, 2)
    model = torch.nn.ModuleDict(dict(news_encoder=head, user_encoder=tail))
    flat = FlatParams(model)
    dp.broadcast_parameters(flat.flat)
    p0 = flat.flat.clone()
    ex = dp.GradientExchange(flat, early_modules=[tail])
    per_gpu, global_batch = shard_sizes(a.batch_size, world, a.weak)
    t0 = time.perf_counter()
    for step in range(a.steps):
        flat.zero_grad()
        flat.grad += float(rank + 1) * (step + 1)                # rank r contributes (r + 1) * (step + 1) to every element
        ex.early_ready()                                         # the tail bucket is final: its all-reduce may start
        scale = ex.finish()
        flat.flat -= 0.5 * scale * flat.grad
    dt = time.perf_counter() - t0
    mean = sum(r + 1 for r in range(world)) / world
    expect = p0 - 0.5 * mean * sum(s + 1 for s in range(a.steps))
    ok = bool(torch.allclose(flat.flat, expect, rtol=0, atol=1e-5))
    same = [torch.zeros_like(flat.flat) for _ in range(world)]
    if world > 1:
        torch.distributed.all_gather(same, flat.flat)
        ok = ok and all(torch.equal(same[0], s) for s in same)
    if rank == 0:
        print(json.dumps({'metric': 'plumbing check (CPU, gloo)', 'n_gpus': world, 'steps': a.steps, 'per_gpu_batch': per_gpu,
                          'global_batch': global_batch, 'scaling': 'weak' if a.weak else 'strong',
                          'buckets': ex.describe(), 'params_equal_on_all_ranks_and_expected': ok, 'seconds': round(dt, 4)}))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0 if ok else 4


def matrix_path(ops, seen=None, classes=None):
    """`config.matrix_path`: which matrix instructions the step's products run on (`dtype` stays "f32": fp32 operands in, fp32 results out,
    fp32 accumulation; the split is exact and the error against fp64 a third of the fp32-MFMA kernel's -- DESIGN.md section 9.4)."""
    if classes is None:
        classes = ops._BX3_CLASSES
    if not ops.BX3[0] or not classes:
        return {'nt_weight_gemms': 'v_mfma_f32_16x16x4_f32', 'weight_gradient_gemms': 'v_mfma_f32_16x16x4_f32', 'recurrences': 'v_mfma_f32_16x16x4_f32 / 4x4x1',
                'switch': 'NNR_BX3=0'}
    out = {'nt_weight_gemms': 'bf16x3: each fp32 operand as three exact bf16 images, six v_mfma_f32_16x16x32_bf16 products, two fp32 accumulators '
                              '(csrc/gemm.hip: gemm_nt_bx3_kernel); shape classes %s, rows >= %d' % (','.join(sorted(classes)), ops._BX3_MIN_ROWS),
           'weight_gradient_gemms': 'v_mfma_f32_16x16x4_f32', 'recurrences': 'v_mfma_f32_16x16x4_f32 / 4x4x1', 'switch': 'NNR_BX3=1 (default)',
           'pure_f32_leg': 'secondary.f32_mfma_only_cne_sue_b64'}
    if seen:
        # eager + recorded steps only (replays do not pass through ops.gemm): which NT shapes took the path ('weight') and which met every
        # other condition but multiply by an activation ('other': left on the fp32 pipe)
        out['launch_classes'] = {'%dx%dx%d %s' % k: v for k, v in sorted(seen.items())}
    return out


def scaling_ceiling(per_gpu, global_batch):
    """N > 1: the ceiling the per-GPU batch sweep of the SAME build puts on this run before any exchange cost -- a rank's step cannot be
    shorter than the 1-GPU step at its per-GPU batch (profiles/batch_sweep.json, written by tools/collect_round6.sh; build-id stamped)."""
    try:
        from nnr_amd import _lib
        d = json.load(open(os.path.join(ROOT, 'profiles', 'batch_sweep.json')))
        have, now = d.get('build_id') or {}, _lib.build_id()
        same = have.get('src_sha256') == now['src_sha256'] or bool(have.get('lib_sha256') and have.get('lib_sha256') == now['lib_sha256'])
        ms = d['ms_per_step'].get(str(per_gpu))
        if ms is None:
            return {'note': 'no 1-GPU measurement at per-GPU batch %d in profiles/batch_sweep.json' % per_gpu}
        return {'per_gpu_batch': per_gpu, 'one_gpu_ms_per_step_at_that_batch': ms, 'value_ceiling': round(global_batch / ms * 1000.0, 1), 'unit': 'impressions/s',
                'how': 'global batch / (1-GPU ms per step at the per-GPU batch): no exchange, no straggler; the measured value cannot exceed it',
                'same_build': bool(same), 'source': 'profiles/batch_sweep.json'}
    except (OSError, ValueError, KeyError):
        return None


def launch_path(trainer):
    if trainer.tapes:
        info = next(iter(trainer.tapes.values())).info()
        return {'path': 'native replay of a recorded launch sequence (nnr_amd/tape.py, csrc/tape.hip)', **info}
    return {'path': trainer.last_path}


def measure_exchange(trainer, torch, dev, world, a):
    """N > 1: what the gradient exchange costs.  Per bucket: bytes and the all-reduce's bus bandwidth measured alone (10 launches
    between HIP events; bus GB/s = bytes x 2 (N - 1) / N / time, the figure a ring moves per link), against the GPU's xGMI
    capacity of 7 links x 153 GB/s; `exposed_ms`: the time the step's main stream spends in GradientExchange.finish() -- waiting for the overlapped
    buckets and reducing the late one -- i.e. the part of the exchange that is NOT hidden behind the backward pass."""
    import torch.distributed as dist
    try:
        ex = trainer.exchange
        out = {'rccl_ranks': dist.get_world_size() if dist.get_backend() == 'nccl' else 0, 'backend': dist.get_backend(), 'binding': ex.describe()['binding'],
               'xgmi_peak_gb_s': 7 * 153, 'buckets': []}
        spans = [('early (user encoder)', ex.early_span), ('table (word embedding)', ex.table_span)] + [('late', s) for s in ex.late_spans]
        scratch = torch.empty_like(trainer.flat.grad)
        for name, span in spans:
            if span is None:
                continue
            view = scratch[span[0]:span[1]]
            for _ in range(2):
                ex._reduce(view, False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dist.barrier()
            e0.record()
            for _ in range(10):
                ex._reduce(view, False)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            nbytes = 4 * (span[1] - span[0])
            out['buckets'].append({'name': name, 'bytes': nbytes, 'allreduce_ms_alone': round(ms, 4),
                                   'bus_gb_s': round(nbytes * 2 * (world - 1) / world / (ms * 1e-3) / 1e9, 1)})
        tot = sum(b['bytes'] for b in out['buckets'])
        tms = sum(b['allreduce_ms_alone'] for b in out['buckets'])
        out['total_bytes'], out['allreduce_ms_alone_total'] = tot, round(tms, 4)
        out['bus_gb_s_overall'] = round(tot * 2 * (world - 1) / world / (tms * 1e-3) / 1e9, 1) if tms > 0 else None
        out['frac_of_xgmi_peak'] = round(out['bus_gb_s_overall'] / (7 * 153), 3) if out['bus_gb_s_overall'] else None
        out['exposed_ms'] = ex.exposed_ms()
        # the table bucket in BOTH forms (round-4 verdict, item 6c): dense [V, E] vs touched rows (flags [V] + packed [U, E]); which one
        # the steps above used is decided by rule (per-GPU batch <= 16 and world > 1, dp.GradientExchange.begin_step)
        if ex.table_span is not None and ex.table_shape is not None:
            V, E = ex.table_shape
            tb = {'rule': ex.describe().get('table_bucket_rule'), 'used': 'touched rows' if ex.touched else 'dense', 'dense_bytes': 4 * V * E}
            if ex.last_touched is not None:
                U = ex.last_touched[0]
                tb.update(touched_rows_last_step=U, touched_bytes=4 * (V + U * E))
            else:
                tb.update(touched_rows_last_step=None, touched_bytes=None,
                          note='the touched-row form did not run in this job (per-GPU batch above the rule\'s threshold); NNR_DP_TOUCHED_ROWS=1 forces it')
            out['table_bucket'] = tb
        return out
    except Exception as e:                      # a measurement of its own: never take the headline line down with it
        return {'error': repr(e)}


def timed_run(a, trainer, fresh, steps, warmup, prof, dp, torch, dev, world, prime_timing):
    """W untimed + K timed steps bracketed by barrier + synchronize; returns the MAX over ranks of the elapsed seconds.  The window is
    UN-INSTRUMENTED (no HIP event pair around any call of any timed step; round-5 verdict: two of the ten steps of a secondary leg carried
    per-call events and made the batch-8 leg 15 % pessimistic); the roofline's per-call durations come from instrumented_steps() AFTER it."""
    for i in range(warmup):
        # prime_timing: the last warm-up step is a TIMING replay whose spans are thrown away -- the first one of a process pays the HIP
        # runtime's lazy set-up of timed events (5-12 ms on a fresh box), which must not land in the instrumented steps after the window
        trainer.timing = bool(prime_timing and i == warmup - 1 and trainer.tapes)
        trainer.train_step(fresh(i))
    trainer.timing = False
    if prime_timing:
        trainer.collect_timings()                    # (discarded: prof.enable() in instrumented_steps starts from empty records)
    dp.barrier()
    torch.cuda.synchronize()
    from nnr_amd import _lib
    calls0 = _lib.CALLS[0]
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if os.environ.get('NNR_BENCH_STEP_MARKS') == '1' else None
    if marks:
        marks[0].record()
    t0 = time.perf_counter()
    host = []
    for i in range(steps):
        trainer.train_step(fresh(warmup + i))
        if marks:
            marks[i + 1].record()
            host.append(time.perf_counter() - t0)
    dp.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if marks:         # diagnostic: where inside the window the time went (one event per step on the main stream; stderr only)
        print('step marks (ms): ' + ' '.join('%.2f' % marks[i].elapsed_time(marks[i + 1]) for i in range(steps)), file=sys.stderr)
        # ... and when the HOST had finished enqueueing each step (ms since the window opened): far below the marks' running sum = the host runs ahead
        print('host enqueue done (ms): ' + ' '.join('%.2f' % (1000 * h) for h in host), file=sys.stderr)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    return float(tmax), (_lib.CALLS[0] - calls0) / max(1, steps)


def instrumented_steps(trainer, fresh, n, first, prof, torch):
    """n extra steps AFTER a timed window, each with a HIP event pair on the launch stream around every GEMM / recurrence / HBM-bound call
    (replayed steps: recorded natively by the tape around the same launches; call-by-call steps: torch events): the live durations behind
    `roofline`.  Same resident batches as the window (fresh(first + i)).  Returns n."""
    from nnr_amd import step as native_step
    taped = bool(trainer.native and trainer.replay and native_step.supported(trainer.model) and trainer.tapes)
    prof.enable(every=1, eager=not taped)
    for tape in trainer.tapes.values():              # (their HIP events exist before the steps: no hipEventCreate between the launches)
        tape.prepare_timing(n)
    for i in range(n):
        trainer.timing = prof.begin_step(i) and taped
        trainer.train_step(fresh(first + i))
    torch.cuda.synchronize()
    trainer.timing = False
    trainer.collect_timings()
    prof.disable()
    return n


SECONDARY_LEGS = (
    # name, encoder pair, dataset, global batch / world of the BASELINE config, per-GPU batch, vocabulary
    ('mhsa_mhsa_b64', 'MHSA', 'MHSA', '200k', 64, 1, 64, 60000),                       # BASELINE.json configs[1]
    ('cne_sue_shard_b8', 'CNE', 'SUE', '200k', 64, 8, 8, 60000),                       # configs[3]: one GPU's shard of batch 64 over 8 GPUs
    ('cne_sue_large_shard_b16_v130000', 'CNE', 'SUE', 'large', 128, 8, 16, 130000),    # configs[4]: MIND-large, batch 128 over 8 GPUs
)
# The headline workload on the PURE fp32-MFMA matrix path (NNR_BX3=0: no launch on the BF16 pipe), driver-timed beside the headline whose
# weight-operand NT GEMMs run as six exact bf16 products with fp32 accumulation (`config.matrix_path`; round-5 verdict, item 1)
F32_ONLY_LEGS = (
    ('f32_mfma_only_cne_sue_b64', 'CNE', 'SUE', '200k', 64, 1, 64, 60000),
)


def secondary_legs(a, prof, dp, torch, dev, rank_seed=0):
    """N = 1: every other BASELINE.json config the GPU can run alone, each as a short driver-run leg on a FRESH model + trainer
    (round-4 verdict, item 3): K = --secondary_steps native replays after --secondary_warmup steps, a fresh batch from the device-
    resident corpus inside every step, the same barrier + synchronize bracket as the headline window.  Per leg: ms_per_step, value
    (impressions/s of that per-GPU shard), step.frac (algorithmic GEMM + recurrence FLOPs of a step / step time / fp32 MFMA peak) and
    the dominant kernel family; the MHSA leg adds `roofline_mhsa` (north_star's MFMA figure of the QK^T / PV contraction).  The
    configs of BASELINE.json that need 8 GPUs appear as the PER-GPU SHARD they put on one MI355X (`shard_of`)."""
    import argparse
    import gc
    import numpy as np
    from nnr_amd.config import make_config
    from nnr_amd.corpus import from_synth
    from nnr_amd.model import Model
    from nnr_amd.synth import SynthSpec, SynthCorpus
    from nnr_amd.trainer import Trainer
    out = {}
    from nnr_amd import ops as _ops, step as native_step
    for name, ne, ue, dataset, gbatch, gworld, per_gpu, V in SECONDARY_LEGS + (() if a.no_experimental else F32_ONLY_LEGS):
        t_leg = time.perf_counter()
        f32_only = name.startswith('f32_mfma_only_')
        bx3_before = _ops.BX3[0]
        _ops.BX3[0] = False if f32_only else bx3_before
        _ops.BX3_SEEN.clear()
        try:
            cfg = make_config(['--news_encoder=' + ne, '--user_encoder=' + ue, '--dataset=' + dataset, '--batch_size=%d' % gbatch,
                               '--world_size=%d' % gworld], corpus_sizes=dict(vocabulary_size=V))
            spec = SynthSpec(vocabulary_size=V)
            torch.manual_seed(cfg.seed)
            table = torch.randn(V, cfg.word_embedding_dim) * 0.3
            table[0] = 0
            model = Model(cfg, table)
            model.initialize()
            trainer = Trainer(model.to(dev).train(), cfg)
            # the SAME draws as a stand-alone run of this shard makes (`bench.py --batch_size <per_gpu> --steps <secondary_steps> --warmup <secondary_warmup>`:
            # rng(100 + rank), 4 096 behaviours, steps + warmup id sets): the leg and that command time the same batches (round-5 verdict, item 4 i)
            rng = np.random.default_rng(100 + rank_seed)
            steps, warm = a.secondary_steps, a.secondary_warmup
            dc = from_synth(SynthCorpus(spec), 4096, rng, dev, graph='build')
            order = [torch.from_numpy(rng.permutation(4096)[:per_gpu].astype(np.int32)).to(dev) for _ in range(steps + warm)]
            src = lambda i: dc.train_batch(order[i % len(order)])
            dt, calls = timed_run(a, trainer, src, steps, warm, prof, dp, torch, dev, 1, True)
            sampled = instrumented_steps(trainer, src, max(1, a.roofline_steps), warm + steps, prof, torch)
            roof = prof.roofline(PEAK_F32_TFLOPS, sampled_steps=sampled, ms_per_step=1000 * dt / steps) or {}
            for v in (roof.get('hbm') or {}).values():      # (the committed PMC passes are of the headline command only)
                v['traffic'] = v['traffic_over_algorithmic'] = None
            leg = {'config': '%s+%s, --dataset=%s, dropout %.2f, V %d' % (ne, ue, dataset, cfg.dropout_rate, V),
                   'shard_of': None if gworld == 1 else {'global_batch': gbatch, 'gpus': gworld},
                   'per_gpu_batch': per_gpu, 'steps': steps, 'warmup': warm, 'ms_per_step': round(1000 * dt / steps, 3),
                   'value': round(steps * per_gpu / dt, 2), 'unit': 'impressions/s (this GPU\'s shard)',
                   'timed_window': 'un-instrumented replays; per-call events from %d extra steps after it' % sampled,
                   'stand_alone': ('`bench.py %s--steps %d --warmup %d --no_secondary` times the SAME batches in a fresh process; the latency-bound small-batch shards run '
                                   '~7 %% faster there (3.10 vs 3.37 ms at batch 8 on one box): which of the step\'s 5-6 HIP streams share one of HIP\'s 4 hardware queues '
                                   'depends on the streams the process used before (profiles/r06_ab.txt calls 40-46)') % (
                                       '--config mhsa ' if ne == 'MHSA' else '--batch_size %d%s ' % (per_gpu, '' if V == 60000 else ' --vocabulary_size %d' % V), steps, warm),
                   'matrix_path': matrix_path(_ops, classes=native_step.bx3_classes(model, per_gpu * (cfg.negative_sample_num + 1 + cfg.max_history_num)))['nt_weight_gemms'],
                   'step': roof.get('step'), 'abi_calls_per_step': round(calls, 1), 'launch_path': launch_path(trainer)['path'],
                   'dominant': {k: roof.get(k) for k in ('kernel', 'family', 'achieved', 'frac', 'avg_launch_us', 'launches', 'share_of_instrumented_time')} if roof else None}
            # the HBM-bound families of THIS leg against the 8 TB/s peak (configs[4] = the "large-vocab embedding table, HBM-bound gather
            # stress": gather / scatter / sumsq / clip_adam at V = 130 000, where the table-proportional traffic is ~1.3 GB per step)
            leg['hbm'] = roof.get('hbm')
            if ne == 'MHSA':
                leg['roofline_mhsa'] = prof.mhsa_roofline(PEAK_F32_TFLOPS)
            for t in list(trainer.tapes.values()):
                t.close()
            trainer.tapes.clear()
            del trainer, model, dc, order
            if f32_only:
                leg['note'] = 'NNR_BX3=0: the headline workload with EVERY matrix product on v_mfma_f32_16x16x4_f32 (the default path of rounds 1-5)'
        except Exception as e:                  # a secondary measurement never takes the headline line down with it
            leg = {'error': repr(e)}
        finally:
            _ops.BX3[0] = bx3_before
        gc.collect()
        torch.cuda.empty_cache()
        leg['leg_seconds'] = round(time.perf_counter() - t_leg, 1)
        out[name] = leg
    return out


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(a))
    if a.plumbing_check:
        sys.exit(plumbing_check(a))

    import numpy as np
    import torch
    from nnr_amd import dp, ops, step as native_step
    from nnr_amd.config import make_config
    from nnr_amd.model import Model
    from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
    from nnr_amd.trainer import Trainer
    from nnr_amd import profile as prof

    # NNR_DP_BACKEND=gloo + NNR_SHARE_GPU=1: test mode -- N ranks on ONE GPU, exchanging through gloo (RCCL refuses two ranks on a device):
    # the launcher, the rank code, the sharding and the bucketed GradientExchange run end to end on device tensors of a 1-GPU box
    rank, local, world = dp.init_from_env(os.environ.get('NNR_DP_BACKEND', 'nccl') if a.gpus > 1 else None)
    if os.environ.get('NNR_SHARE_GPU') == '1':
        local = 0
    assert world == a.gpus, 'world size %d != --gpus %d (launch with torch.distributed.run --nproc-per-node %d, or let bench.py do it)' % (world, a.gpus, a.gpus)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    weak = bool(a.weak)
    per_gpu, global_batch = shard_sizes(a.batch_size, world, weak)
    assert per_gpu >= 1, 'global batch %d cannot be split over %d GPUs' % (a.batch_size, world)
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
    device_corpus = not a.prebuilt
    nb = min(256, a.steps + a.warmup) if device_corpus else min(8, a.steps + a.warmup)
    dcorpus = []

    def batch_source(per_rank):
        if device_corpus:
            # a1's batch delivery inside the step: every step gathers / builds its 21 tensors in HBM from the device-resident corpus
            # (nnr_corpus_batch + nnr_history_graph) out of `per_rank` behaviour ids; steps + warmup DISTINCT id sets, a fresh batch
            # each step (the sustained leg cycles through the same sets: other draws are another workload, +-6 % in tokens)
            from nnr_amd.corpus import from_synth
            if not dcorpus:
                dcorpus.append(from_synth(corpus, 4096, rng, dev, graph='build'))
            order = [torch.from_numpy(rng.permutation(4096)[:per_rank].astype(np.int32)).to(dev) for _ in range(nb)]
            return lambda i: dcorpus[0].train_batch(order[i % nb])
        batches = [to_torch(corpus.batch(per_rank, rng), dev) for _ in range(nb)]
        return lambda i: batches[i % nb]        # masks are mutated in place by the model; mutation is idempotent, so batches can be reused

    headline_batches = batch_source(per_gpu)      # the SAME resident batches serve the timed region, the sustained leg and the isolated leg (a
                                                  # MIND-shaped batch of 64 varies by +-6 % in tokens: other draws are another workload)
    dt, calls = timed_run(a, trainer, headline_batches, a.steps, a.warmup, prof, dp, torch, dev, world, True)
    bx3_seen_headline = dict(ops.BX3_SEEN)
    sampled = instrumented_steps(trainer, headline_batches, max(1, a.roofline_steps), a.warmup + a.steps, prof, torch)
    roof = prof.roofline(PEAK_F32_TFLOPS, sampled_steps=sampled, ms_per_step=1000 * dt / a.steps)
    if roof:
        roof['timed_window'] = 'un-instrumented; per-call HIP events from %d extra steps after the window (same resident batches)' % sampled
        # (the committed kernel tables are of the HEADLINE command: other batch sizes / encoder pairs launch the same kernels at other sizes)
        roof['rocprof'] = prof.rocprof_block(roof, PEAK_F32_TFLOPS) if (a.news_encoder, a.user_encoder, per_gpu, a.dense, world) == ('CNE', 'SUE', 64, False, 1) else None
        if str(roof.get('family', '')).startswith('gemm_nt_bx3'):
            # fp32-EQUIVALENT FLOPs (2 M N K) against the fp32 MFMA peak, as for every other family -- the kernel itself issues six bf16 MFMA products per
            # fp32 product, so ITS matrix-pipe bound is the dense bf16 rate / 6
            roof['peak_note'] = ('achieved / frac are fp32-equivalent FLOPs (2 M N K) against the fp32 MFMA peak of %.1f TFLOP/s; the bf16x3 kernel runs six '
                                 'v_mfma_f32_16x16x32_bf16 products per fp32 product: its own matrix-pipe bound is 2 500 / 6 = 416.7 TFLOP/s-equivalent' % PEAK_F32_TFLOPS)
            roof['frac_of_bf16x3_bound'] = round(roof['achieved'] / (2500.0 / 6.0), 4)
    headline = (a.news_encoder, a.user_encoder, per_gpu, a.dense) == ('CNE', 'SUE', 64, False)
    if roof and not headline:
        roof['traffic'] = None            # the committed PMC passes (profiles/pmc_traffic.json) are of the headline command only
        for v in (roof.get('hbm') or {}).values():
            v['traffic'] = v['traffic_over_algorithmic'] = None
    if roof and a.news_encoder == 'MHSA':
        roof['mhsa'] = prof.mhsa_roofline(PEAK_F32_TFLOPS)
    if roof and headline:
        # counter bytes (same PMC passes) over ALGORITHMIC operand bytes of the weight-gradient (token-reduction) GEMM launches
        roof['weight_gradient_traffic'] = prof.weight_gradient_traffic()
    if roof and world == 1 and not a.no_isolated:
        # The dominant kernel's launches overlap with up to three other HIP streams inside the step, so `achieved` above divides its
        # FLOPs by a wall duration it shares with them.  Two extra untimed steps with every launch serialised on ONE stream give the
        # solo duration of the same launches at the same live sizes (the figure that measures the kernel, not the schedule).
        fresh = headline_batches
        ops.set_one_stream(True)
        for i in range(2):                   # (two warm steps: the one-stream order allocates its ~10 GB of temporaries afresh; on some
            trainer.train_step(fresh(i))     # boxes the first such steps took 150 ms each)
        torch.cuda.synchronize()
        prof.enable(every=1)
        sers = []
        for i in range(2):
            prof.begin_step(i)
            t0 = time.perf_counter()
            trainer.train_step(fresh(2 + i))
            torch.cuda.synchronize()
            sers.append(time.perf_counter() - t0)
        ser = min(sers)
        fam = prof.summary().get(roof['family'])
        prof.disable()
        ops.set_one_stream(False)
        if fam and fam['ms'] > 0:
            tf = fam['flops'] / (fam['ms'] * 1e-3) / 1e12
            roof['isolated'] = {'achieved': round(tf, 2), 'frac': round(tf / PEAK_F32_TFLOPS, 4), 'avg_launch_us': round(1000 * fam['ms'] / fam['launches'], 2),
                                'launches': fam['launches'], 'step_serialized_ms': round(1000 * ser, 3),
                                'how': 'same launches of 2 extra untimed steps, every HIP stream of the step collapsed into one'}

    sustained = None
    if a.sustained_seconds > 0:
        # the driver's K = 20 steps are 0.2 s of GPU time; the same loop for >= 3 s shows the clocks the chip holds under this load
        n_sus = max(a.steps, int(a.sustained_seconds / max(1e-4, dt / a.steps)) + 1)
        sdt_, _ = timed_run(a, trainer, headline_batches, n_sus, 0, prof, dp, torch, dev, world, False)
        sustained = {'seconds': round(sdt_, 3), 'steps': n_sus, 'ms_per_step': round(1000 * sdt_ / n_sus, 3),
                     'value': round(n_sus * global_batch / sdt_, 2), 'unit': 'impressions/s'}

    exchange = None
    if world > 1:
        exchange = measure_exchange(trainer, torch, dev, world, a)

    other = None
    if world > 1 and not a.no_weak:
        # the OTHER scaling mode of the same job, as a secondary object: weak scaling (batch_size impressions per GPU) beside the
        # strong-scaling headline, or -- under --weak -- the reference's global-batch semantics beside the weak headline
        op_, og_ = shard_sizes(a.batch_size, world, not weak)
        if op_ >= 1 and op_ != per_gpu:
            odt, ocalls = timed_run(a, trainer, batch_source(op_), a.steps, max(3, a.warmup // 2), prof, dp, torch, dev, world, False)
            other = {'scaling': 'strong' if weak else 'weak', 'global_batch': og_, 'per_gpu_batch': op_, 'value': round(a.steps * og_ / odt, 2),
                     'unit': 'impressions/s', 'ms_per_step': round(1000 * odt / a.steps, 3), 'abi_calls_per_step': round(ocalls, 1)}

    secondary = None
    if world == 1 and not a.no_secondary and headline:
        secondary = secondary_legs(a, prof, dp, torch, dev)

    exchange_timeouts = ops.lstm_sync_timeouts()      # persistent device counter over EVERY pair-kernel launch of this process
    tmo = torch.tensor([exchange_timeouts], device=dev, dtype=torch.int64)
    if world > 1:
        torch.distributed.all_reduce(tmo)
    exchange_timeouts = int(tmo)
    if exchange_timeouts and rank == 0:
        print('ERROR: pair-recurrence exchange timed out %d times (values poisoned with NaN, optimizer steps skipped)' % exchange_timeouts,
              file=sys.stderr)

    if rank == 0:
        out = {
            'metric': ('training impressions/sec on MIND-200k (%s+%s, bs=%d)' % (a.news_encoder, a.user_encoder, a.batch_size)) +
                      (' per GPU, weak scaling' if (weak and world > 1) else ''),
            'value': round(a.steps * global_batch / dt, 2), 'unit': 'impressions/s', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': round(1000 * dt / a.steps, 3), 'higher_is_better': True,
            'scaling': 'weak' if weak else 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s+%s train step, MIND-200k-shaped synthetic batches, dropout %.2f, gcn_layer_num %d%s' %
                                   (a.news_encoder, a.user_encoder, cfg.dropout_rate, cfg.gcn_layer_num, ', dense lengths' if a.dense else ''),
                       'global_batch': global_batch, 'per_gpu_batch': per_gpu, 'parallelism': 'dp%d' % world,
                       'matrix_path': matrix_path(ops, bx3_seen_headline, classes=native_step.bx3_classes(trainer.model, per_gpu * (cfg.negative_sample_num + 1 + cfg.max_history_num))),
                       'synth': {k: v for k, v in spec.describe().items() if k in ('vocabulary_size', 'title_len_mean', 'content_len_mean', 'news_pool', 'dense')},
                       'batches': ('device-resident corpus, id-only: a fresh batch is gathered / built in HBM inside every timed step (%d distinct id sets)' % nb)
                                  if device_corpus else 'pre-built, %d batches resident in HBM, re-used' % nb,
                       'peak_hbm_reserved_gb': round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 2),
                       'abi_calls_per_step': round(calls, 1),
                       'launch_path': launch_path(trainer),
                       'gradient_exchange': trainer.exchange.describe() if world > 1 else 'none (1 GPU)',
                       'recurrence_exchange_timeouts': exchange_timeouts},
            'roofline': roof,
        }
        if sustained is not None:
            out['sustained'] = sustained
        if exchange is not None:
            out['exchange'] = exchange
        if world > 1:
            out['scaling_ceiling'] = scaling_ceiling(per_gpu, global_batch)
        if other is not None:
            out['weak_scaling' if other['scaling'] == 'weak' else 'strong_scaling'] = other
        if secondary is not None:
            out['secondary'] = secondary
        if not a.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(cfg, spec, a.cpu_baseline_batch, a.cpu_baseline_steps,
                                               headline_batch=per_gpu if a.cpu_baseline_headline_steps > 0 else 0, headline_steps=a.cpu_baseline_headline_steps,
                                               headline_threads=a.cpu_baseline_headline_threads)
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        torch.distributed.destroy_process_group()
    if exchange_timeouts:
        sys.exit(3)


if __name__ == '__main__':
    main()
