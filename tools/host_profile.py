#!/usr/bin/env python3
"""cProfile of the host side of the training step at a small batch (where the step is bound by Python's launch rate).
Usage: host_profile.py [batch]"""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nnr_amd.config import make_config
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda')
cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=%d' % B], corpus_sizes=dict(vocabulary_size=60000))
torch.manual_seed(1)
table = torch.randn(cfg.vocabulary_size, cfg.word_embedding_dim) * 0.3
model = Model(cfg, table); model.initialize(); model = model.to(dev).train()
tr = Trainer(model, cfg)
corpus = SynthCorpus(SynthSpec(vocabulary_size=60000))
rng = np.random.default_rng(0)
batches = [to_torch(corpus.batch(B, rng), dev) for _ in range(4)]
for i in range(6):
    tr.train_step(batches[i % 4])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    tr.train_step(batches[i % 4])
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats('tottime')
ps.print_stats(28)
print(s.getvalue()[:6000])
