#!/usr/bin/env python3
"""Undistorted GPU timeline of one train step: every C-ABI call is bracketed by two HIP events on its launch stream (about
+3 us of host time per call: the host stays far ahead of the GPU, unlike under rocprofv3's tracer, which makes the step
host-bound).  Prints the calls of one step in start order with stream, start, duration, and the idle gap before each call on
its stream; then a summary of which stream was the only one busy, per phase.
Usage: event_timeline.py [batch] [min_us]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnr_amd import _lib as L
from nnr_amd import ops
from nnr_amd.config import make_config
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
MIN_US = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dev = torch.device('cuda')
cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=%d' % B], corpus_sizes=dict(vocabulary_size=60000))
torch.manual_seed(0)
model = Model(cfg, torch.randn(cfg.vocabulary_size, cfg.word_embedding_dim) * 0.3); model.initialize(); model = model.to(dev).train()
tr = Trainer(model, cfg)
corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size))
rng = np.random.default_rng(100)
batches = [to_torch(corpus.batch(B, rng), dev) for _ in range(4)]
for i in range(6):
    tr.train_step(batches[i % 4])
torch.cuda.synchronize()

real = L.lib()
rec = []
on = [False]


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if not name.startswith('nnr_') or name in ('nnr_lstm_dims', 'nnr_lstm_sync_bytes', 'nnr_lstm_sync_diag_offset', 'nnr_slot_workspace_floats', 'nnr_version'):
            return fn

        def wrapped(*a):
            if not on[0]:
                return fn(*a)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = torch.cuda.current_stream()
            s.record(st)
            r = fn(*a)
            e.record(st)
            tag = ''
            if name == 'nnr_gemm_f32':
                g = a[0]._obj
                tag = ' %dx%dx%d%s%s' % (g.M, g.N, g.K, ' TN' if g.trans_a else (' NN' if g.trans_b else ''), ' sk%d' % g.split_k if g.split_k > 1 else '')
            rec.append((name[4:] + tag, st.cuda_stream, s, e))
            return r
        return wrapped


L._lib = Proxy()
origin = torch.cuda.Event(enable_timing=True)
steps = []
on[0] = True
for i in range(3):
    mark = len(rec)
    o = torch.cuda.Event(enable_timing=True); o.record()
    h0 = time.perf_counter()
    tr.train_step(batches[i % 4])
    steps.append((mark, len(rec), o, time.perf_counter() - h0))
on[0] = False
endev = torch.cuda.Event(enable_timing=True); endev.record()
torch.cuda.synchronize()
a, b, o, host = steps[-1]
print('batch %d: host enqueue of the instrumented step %.2f ms, GPU step %.3f ms, %d C-ABI calls' % (B, host * 1e3, o.elapsed_time(endev), b - a))
streams = {}
rows = []
for name, st, s, e in rec[a:b]:
    sid = streams.setdefault(st, len(streams))
    rows.append((o.elapsed_time(s) * 1e3, o.elapsed_time(e) * 1e3, sid, name))
rows.sort()
last_end = collections.defaultdict(float)
for s, e, sid, name in rows:
    gap = s - last_end[sid]
    last_end[sid] = e
    if e - s >= MIN_US or gap >= 50:
        print('%9.1f %8.1f  s%d  gap %7.1f  %s' % (s, e - s, sid, gap, name))
# exclusive-time accounting: sweep
evs = []
for s, e, sid, name in rows:
    evs.append((s, 1, sid)); evs.append((e, -1, sid))
evs.sort()
active = collections.Counter(); prev = 0.0; excl = collections.Counter(); idle = 0.0; multi = 0.0
for t, d, sid in evs:
    n = sum(1 for v in active.values() if v > 0)
    if n == 0: idle += t - prev
    elif n == 1: excl[[k for k, v in active.items() if v > 0][0]] += t - prev
    else: multi += t - prev
    active[sid] += d; prev = t
print('no call in flight %.3f ms; exactly one stream busy: %s ms; >= 2 streams busy %.3f ms' % (idle / 1e3, {('s%d' % k): round(v / 1e3, 3) for k, v in excl.items()}, multi / 1e3))
