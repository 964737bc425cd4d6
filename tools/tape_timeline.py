#!/usr/bin/env python3
"""Per-call GPU timeline of ONE replayed training step (the step bench.py times): every recorded C-ABI call carries a HIP event pair
on its own stream (NNR tape timing replay with all calls tagged).  Usage: python tools/tape_timeline.py [--batch_size 64]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nnr_amd import tape as T
from nnr_amd.config import make_config
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer

ap = argparse.ArgumentParser()
ap.add_argument('--batch_size', type=int, default=64)
ap.add_argument('--vocabulary_size', type=int, default=60000)
ap.add_argument('--news_encoder', default='CNE')
ap.add_argument('--user_encoder', default='SUE')
a = ap.parse_args()
T.TAG_ALL[0] = True
cfg = make_config(['--news_encoder=' + a.news_encoder, '--user_encoder=' + a.user_encoder, '--dataset=200k', '--batch_size=%d' % a.batch_size], corpus_sizes=dict(vocabulary_size=a.vocabulary_size))
torch.manual_seed(0)
table = torch.randn(cfg.vocabulary_size, cfg.word_embedding_dim) * 0.3
table[0] = 0
model = Model(cfg, table)
model.initialize()
tr = Trainer(model.cuda().train(), cfg)
corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size))
rng = np.random.default_rng(100)
batches = [to_torch(corpus.batch(a.batch_size, rng), 'cuda') for _ in range(4)]
for i in range(6):
    tr.train_step(batches[i % 4])
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(10):
    tr.train_step(batches[i % 4])
torch.cuda.synchronize()
plain = (time.perf_counter() - t0) / 10
tr.timing = True
tr.train_step(batches[0])
tr.timing = False
torch.cuda.synchronize()
tape = next(iter(tr.tapes.values()))
rows = tape.timeline(0)
end = max(s + d for s, d, *_ in rows)
print('batch %d: replayed step %.3f ms untimed (10 steps), %.3f ms under per-call events; %s' % (a.batch_size, 1000 * plain, end, tape.info()))
prev_end = {}
for s, d, st, fam, tag in sorted(rows):
    gap = s - prev_end.get(st, 0.0)
    prev_end[st] = s + d
    print('%9.1f %8.1f  s%d  gap %7.1f  %s %s' % (1000 * s, 1000 * d, st, 1000 * gap, fam, tag))
# busy statistics
ev = sorted([(s, 1) for s, d, *_ in rows] + [(s + d, -1) for s, d, *_ in rows])
busy = {0: 0.0, 1: 0.0, 2: 0.0}
lvl, last = 0, 0.0
for t, k in ev:
    busy[min(lvl, 2)] += t - last
    lvl += k
    last = t
print('no call in flight %.3f ms; exactly one %.3f ms; >= 2 in flight %.3f ms' % (busy[0], busy[1], busy[2]))
