import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch threads default', torch.get_num_threads(), flush=True)
from nnr_amd.config import make_config
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from oracle import nnr_oracle as O
cfg = make_config([], corpus_sizes=dict(vocabulary_size=60000))
spec = SynthSpec(vocabulary_size=60000)
corpus = SynthCorpus(spec)
for nt in (8, 16, 32):
    torch.set_num_threads(nt)
    torch.manual_seed(0)
    m = O.Model(cfg); m.initialize(); m.train()
    opt = O.make_optimizer(m, cfg)
    b = to_torch(corpus.batch(4, np.random.default_rng(1)))
    t0 = time.perf_counter(); O.train_step(m, opt, b, 4.0); t1 = time.perf_counter()
    b = to_torch(corpus.batch(4, np.random.default_rng(2)))
    O.train_step(m, opt, b, 4.0); t2 = time.perf_counter()
    print('threads', nt, 'bs4 step: first %.1fs second %.1fs' % (t1 - t0, t2 - t1), flush=True)
