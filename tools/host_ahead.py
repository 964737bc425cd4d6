#!/usr/bin/env python3
"""How far ahead of the GPU is the host at the phase boundaries of a train step (no tracer)?  Host clock at the moment a
boundary is ENQUEUED vs the GPU time (HIP event on the main stream) at which the GPU REACHES it, both from a common origin."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnr_amd import ops
from nnr_amd.config import make_config
from nnr_amd.model import Model, negative_log_softmax
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
N = 8
marks = []          # (name, host_t, event)
origin = torch.cuda.Event(enable_timing=True)
h0 = time.perf_counter(); origin.record()


def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record()
    marks.append((name, time.perf_counter() - h0, e))


for i in range(N):
    b = batches[i % 4]
    mark('step start')
    tr.flat.zero_grad()
    logits = model(*b)
    loss = negative_log_softmax(logits)
    mark('forward enqueued')
    loss.backward()
    mark('backward enqueued')
    ops.join_extra_streams()
    tr.optimizer_step(tr.exchange.finish())
    mark('optimizer enqueued')
torch.cuda.synchronize()
print('batch %d' % B)
per = len(marks) // N
for i in range(N - 3, N):
    for name, ht, e in marks[i * per:(i + 1) * per]:
        gt = origin.elapsed_time(e)
        print('  step %d %-20s host %8.3f ms   gpu %8.3f ms   host ahead by %7.3f ms' % (i, name, ht * 1e3, gt, gt - ht * 1e3))
