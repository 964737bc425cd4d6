#!/usr/bin/env python3
"""Host time of one native replay (the C loop of entry-point calls + event ops) vs the GPU time of the step."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nnr_amd.config import make_config
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer

ap = argparse.ArgumentParser()
ap.add_argument('--batch_size', type=int, default=8)
a = ap.parse_args()
cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=%d' % a.batch_size])
torch.manual_seed(0)
model = Model(cfg); model.initialize()
tr = Trainer(model.cuda().train(), cfg)
corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size))
rng = np.random.default_rng(100)
batches = [to_torch(corpus.batch(a.batch_size, rng), 'cuda') for _ in range(4)]
for i in range(6):
    tr.train_step(batches[i % 4])
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for i in range(50):
    h0 = time.perf_counter()
    tr.train_step(batches[i % 4])
    host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 50
print('batch %d: step %.3f ms; host time inside train_step (replay, no sync): median %.3f ms, min %.3f, max %.3f; path %s' %
      (a.batch_size, 1000 * total, 1000 * float(np.median(host)), 1000 * min(host), 1000 * max(host), tr.last_path))
# a single replay from an idle GPU: host enqueue vs GPU completion
for _ in range(3):
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    tr.train_step(batches[0])
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    h2 = time.perf_counter()
    print('  from idle: enqueue %.3f ms, completion %.3f ms after the call started' % (1000 * (h1 - h0), 1000 * (h2 - h0)))
