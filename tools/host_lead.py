#!/usr/bin/env python3
"""Is the host ahead of the GPU when the CNE backward forks?  Two HIP events on the main stream around the HOST-side issue of
the candidate call's backward-pre phase (nothing is enqueued on the main stream between them): if the host runs ahead of the
GPU both are processed back to back (elapsed ~ 0); an elapsed time of hundreds of us means the main stream sat idle waiting for
the host to issue the history call's launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnr_amd import news_encoders as NE, ops
from nnr_amd.config import make_config
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer

BS = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=%d' % BS], corpus_sizes=dict(vocabulary_size=60000))
torch.manual_seed(0)
model = Model(cfg, torch.randn(cfg.vocabulary_size, cfg.word_embedding_dim) * 0.3)
model.initialize()
model = model.cuda().train()
tr = Trainer(model, cfg)
corpus = SynthCorpus(SynthSpec(vocabulary_size=60000))
rng = np.random.default_rng(0)
batches = [to_torch(corpus.batch(BS, rng), 'cuda') for _ in range(4)]
pairs = []
orig = NE._fork_join


def patched(n_calls, dev, phase):
    if n_calls == 1:
        return orig(n_calls, dev, phase)
    main = torch.cuda.current_stream(dev)
    side = NE._side_stream(dev)
    side.wait_stream(main)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    with torch.cuda.stream(side):
        first = phase(0, False)
    e1.record(main)
    rest = [phase(i, True) for i in range(1, n_calls)]
    main.wait_stream(side)
    pairs.append((e0, e1))
    return [first] + rest


NE._fork_join = patched
for i in range(16):
    if i == 6:
        pairs.clear()
    tr.train_step(batches[i % 4])
torch.cuda.synchronize()
# per step: forward pre, forward post, backward pre (3 forks)
names = ['forward pre', 'forward post', 'backward pre']
el = np.array([a.elapsed_time(b) for a, b in pairs]).reshape(-1, 3)
for k, n in enumerate(names):
    print('%-13s main-stream idle while the host issues the candidate phase: mean %.3f ms, max %.3f ms' % (n, el[:, k].mean(), el[:, k].max()))
