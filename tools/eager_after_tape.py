#!/usr/bin/env python3
"""Diagnostics: per-step wall time of call-by-call native steps issued AFTER a tape was recorded (the tape keeps ~11 GB of buffers at
batch 64), multi-stream and with every stream collapsed into one (bench.py's `isolated` leg)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nnr_amd import ops
from nnr_amd.config import make_config
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer

cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=64'])
torch.manual_seed(0)
model = Model(cfg); model.initialize()
tr = Trainer(model.cuda().train(), cfg)
corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size))
rng = np.random.default_rng(100)
batches = [to_torch(corpus.batch(64, rng), 'cuda') for _ in range(4)]


def run(n, label):
    ts = []
    for i in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.train_step(batches[i % 4])
        torch.cuda.synchronize(); ts.append(1000 * (time.perf_counter() - t0))
    print('%-34s %s  (path %s, reserved %.1f GB)' % (label, ' '.join('%.1f' % t for t in ts), tr.last_path, torch.cuda.memory_reserved() / 2 ** 30), flush=True)


run(6, 'warm-up / record / replay')
tr.replay = False
run(6, 'call by call after the tape')
ops.set_one_stream(True)
run(5, 'one stream, call by call')
ops.set_one_stream(False)
tr.replay = True
run(4, 'replay again')
# bench.py's isolated leg: eager HIP-event spans on every launch, one stream
from nnr_amd import profile as prof
ops.set_one_stream(True)
run(1, 'one stream (untimed warm step)')
prof.enable(every=1)
for i in range(3):
    prof.begin_step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.train_step(batches[i % 4])
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print('one stream + event spans: enqueue %.1f ms, total %.1f ms (path %s)' % (1000 * (t1 - t0), 1000 * (t2 - t0), tr.last_path), flush=True)
t0 = time.perf_counter()
fam = prof.summary()
print('summary() took %.1f ms, %d families' % (1000 * (time.perf_counter() - t0), len(fam)))
prof.disable()
ops.set_one_stream(False)
