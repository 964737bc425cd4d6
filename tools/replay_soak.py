#!/usr/bin/env python3
"""Soak of the replayed CNE+SUE step at the headline shape: N optimizer steps over 8 resident MIND-shaped batches (dropout on, seeds
advance every step), then: recurrence exchange time-outs (must be 0), Adam steps skipped for a non-finite gradient norm (must be 0),
finite loss / parameters, the loss trend and the step time.  Usage: python tools/replay_soak.py [--steps 1500]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nnr_amd import ops
from nnr_amd.config import make_config
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=1500)
ap.add_argument('--batch_size', type=int, default=64)
ap.add_argument('--busy', type=int, default=0, help='workgroups of a resident ring-kernel stand-in (nnr_dp_busy) kept running on a side stream beside '
                'every step: the CU-pair recurrence beside RCCL-like resident kernels (round-4 verdict, item 6a); 0 = off')
ap.add_argument('--busy_iters', type=int, default=0, help='sweeps per busy launch (0: sized to ~ one step)')
a = ap.parse_args()
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
ops.lstm_sync_timeouts(reset=True)
losses, paths = [], {}
busy_stream = torch.cuda.Stream() if a.busy > 0 else None
busy_buf = torch.ones(a.busy * 512 * 64 * 2, device='cuda') if a.busy > 0 else None      # 256 KB per workgroup
busy_iters = a.busy_iters or (400 if a.batch_size <= 16 else 1200)


def busy():
    if busy_stream is not None:                      # one launch per step, back to back on its own stream: resident the whole time
        with torch.cuda.stream(busy_stream):
            ops.dp_busy(busy_buf, a.busy, busy_iters)

for i in range(8):
    tr.train_step(batches[i % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    busy()
    out = tr.train_step(batches[i % 8])
    paths[tr.last_path] = paths.get(tr.last_path, 0) + 1
    if i % 100 == 0 or i == a.steps - 1:
        losses.append((i, float(out[1])))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
res = {'steps': a.steps, 'batch_size': a.batch_size, 'busy_workgroups': a.busy, 'busy_sweeps_per_launch': busy_iters if a.busy else 0, 'ms_per_step_including_%d_loss_readbacks' % len(losses): round(1000 * dt / a.steps, 3), 'paths': paths,
       'recurrence_exchange_timeouts': int(ops.lstm_sync_timeouts()), 'adam_steps_skipped': int(tr.skipped_steps()), 'parameters_finite': finite,
       'loss_every_100_steps': [(i, round(l, 4)) for i, l in losses]}
print(json.dumps(res))
assert res['recurrence_exchange_timeouts'] == 0 and res['adam_steps_skipped'] == 0 and finite and all(np.isfinite(l) for _, l in losses)
