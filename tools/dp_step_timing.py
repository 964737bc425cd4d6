#!/usr/bin/env python3
"""Step time of ONE rank's shard with the gradient exchange IN the step, on one GPU: a one-rank RCCL communicator (NNR_DP_FORCE=1: every bucket is
exchanged -- a one-rank all-reduce moves no data between GPUs, but the collective's launches, the exchange's extra HIP streams and their hardware
queues are all there).  `--exchange 0` = the same process without it.  Usage: python tools/dp_step_timing.py --batch_size 8 --exchange 1"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--batch_size', type=int, default=8)
ap.add_argument('--exchange', type=int, default=1)
ap.add_argument('--steps', type=int, default=40)
ap.add_argument('--timeline', action='store_true', help='per-call timeline of one replayed step (every call tagged)')
a = ap.parse_args()
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', '29547'))
os.environ['NNR_DP_FORCE'] = '1' if a.exchange else '0'
import numpy as np
import torch
from nnr_amd import tape as T
if a.timeline:
    T.TAG_ALL[0] = True
import torch.distributed as dist
from nnr_amd import dp
from nnr_amd.config import make_config
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer

torch.cuda.set_device(0)
if a.exchange:
    dist.init_process_group('nccl', init_method='env://', world_size=1, rank=0)
cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=%d' % a.batch_size], corpus_sizes=dict(vocabulary_size=60000))
torch.manual_seed(0)
table = torch.randn(cfg.vocabulary_size, cfg.word_embedding_dim) * 0.3
table[0] = 0
model = Model(cfg, table)
model.initialize()
tr = Trainer(model.cuda().train(), cfg)
corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size))
rng = np.random.default_rng(100)
batches = [to_torch(corpus.batch(a.batch_size, rng), 'cuda') for _ in range(8)]
for i in range(8):
    tr.train_step(batches[i % 8])
torch.cuda.synchronize()
best = []
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(a.steps):
        tr.train_step(batches[i % 8])
    torch.cuda.synchronize()
    best.append(1000 * (time.perf_counter() - t0) / a.steps)
print('batch %d exchange %d: %s ms/step; path %s; exchange %s; streams created by the package: %d' % (
    a.batch_size, a.exchange, ' '.join('%.3f' % x for x in best), tr.last_path, tr.exchange.describe() if tr.exchange.active() else 'inactive',
    len(__import__('nnr_amd.ops', fromlist=['x']).EXTRA_STREAMS)))
if a.timeline and tr.tapes:
    tr.timing = True
    tr.train_step(batches[0])
    tr.timing = False
    torch.cuda.synchronize()
    tape = next(iter(tr.tapes.values()))
    print(tape.info())
    prev = {}
    for s_, d_, st, fam, tag in sorted(tape.timeline(0)):
        gap = s_ - prev.get(st, 0.0)
        prev[st] = s_ + d_
        print('%9.1f %8.1f  s%d  gap %7.1f  %s %s' % (1000 * s_, 1000 * d_, st, 1000 * gap, fam, tag))
if a.exchange:
    dist.destroy_process_group()
