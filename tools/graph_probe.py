#!/usr/bin/env python3
"""Feasibility probe: capture one whole train step (forward, loss, backward, clip+Adam on 4 HIP streams) into a HIP graph with
torch.cuda.graph and replay it.  Seeds / Adam step are frozen in this probe (timing and capturability only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnr_amd.config import make_config
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda')
cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=%d' % B], corpus_sizes=dict(vocabulary_size=60000))
torch.manual_seed(0)
table = torch.randn(cfg.vocabulary_size, cfg.word_embedding_dim) * 0.3
model = Model(cfg, table); model.initialize(); model = model.to(dev).train()
trainer = Trainer(model, cfg)
corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size))
rng = np.random.default_rng(100)
batches = [to_torch(corpus.batch(B, rng), dev) for _ in range(4)]
static = [t.clone() for t in batches[0]]


def load(i):
    for s, t in zip(static, batches[i % 4]):
        s.copy_(t)


for i in range(4):
    load(i); trainer.train_step(static)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    load(i); trainer.train_step(static)
torch.cuda.synchronize()
print('eager   %.3f ms/step' % ((time.perf_counter() - t0) * 100))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    load(0); trainer.train_step(static)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        out = trainer.train_step(static)
except Exception as e:
    print('capture failed:', type(e).__name__, str(e)[:2000])
    sys.exit(1)
torch.cuda.synchronize()
print('captured')
for i in range(3):
    load(i); g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    load(i); g.replay()
torch.cuda.synchronize()
print('replay  %.3f ms/step   loss %.5f' % ((time.perf_counter() - t0) * 50, float(out[1])))
