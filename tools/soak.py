#!/usr/bin/env python3
"""Soak: N training steps with dropout on; the loss must stay finite, the pair-recurrence exchange must never time out,
and parameters must stay finite.  (Synthetic clicks carry no signal, so the loss hovers around log 5.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnr_amd import ops
from nnr_amd.config import make_config
from nnr_amd.corpus import from_synth
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus
from nnr_amd.trainer import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
BS = int(sys.argv[2]) if len(sys.argv) > 2 else 64
NE, UE = (sys.argv[3], sys.argv[4]) if len(sys.argv) > 4 else ('CNE', 'SUE')
cfg = make_config(['--news_encoder=' + NE, '--user_encoder=' + UE, '--dataset=200k', '--batch_size=%d' % BS], corpus_sizes=dict(vocabulary_size=60000))
torch.manual_seed(0)
model = Model(cfg, torch.randn(cfg.vocabulary_size, cfg.word_embedding_dim) * 0.3)
model.initialize()
model = model.cuda().train()
tr = Trainer(model, cfg)
synth = SynthCorpus(SynthSpec(vocabulary_size=60000))
rng = np.random.default_rng(0)
dc = from_synth(synth, 8192, rng, 'cuda')
import time
losses, timeouts, times = [], 0, []
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx = torch.from_numpy(rng.permutation(8192)[:BS].astype(np.int32)).cuda()
    _, loss = tr.train_step(dc.train_batch(idx))
    losses.append(loss)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    timeouts += ops.lstm_sync_timeouts()
l = torch.stack(losses).cpu().numpy()
flat = tr.flat.flat
print(NE, UE, 'reserved GB %.2f' % (torch.cuda.max_memory_reserved() / 2 ** 30))
print('batch %d steps %d: loss first %.4f last-50 mean %.4f min %.4f max %.4f; finite loss %s; finite params %s; exchange timeouts %d' %
      (BS, steps, l[0], l[-50:].mean(), l.min(), l.max(), bool(np.isfinite(l).all()), bool(torch.isfinite(flat).all()), timeouts))
ts = np.array(times[5:]) * 1e3
print('step time (sync each step): median %.2f ms, p99 %.2f ms, max %.2f ms' % (np.median(ts), np.percentile(ts, 99), ts.max()))
assert np.isfinite(l).all() and bool(torch.isfinite(flat).all()) and timeouts == 0
