"""How long does the host take to ENQUEUE one train step (no GPU wait)?  Decides whether hipGraph capture pays."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnr_amd.config import make_config
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer

for bs in (2, 8, 64):
    cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=%d' % bs], corpus_sizes=dict(vocabulary_size=60000))
    torch.manual_seed(1)
    table = torch.randn(cfg.vocabulary_size, cfg.word_embedding_dim) * 0.3
    model = Model(cfg, table); model.initialize(); model = model.cuda().train()
    tr = Trainer(model, cfg)
    corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size))
    rng = np.random.default_rng(3)
    bts = [to_torch(corpus.batch(bs, rng), 'cuda') for _ in range(4)]
    for i in range(4):
        tr.train_step(bts[i % 4])
    torch.cuda.synchronize()
    enq, tot = [], []
    for i in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train_step(bts[i % 4])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    t0 = time.perf_counter()
    for i in range(20):
        tr.train_step(bts[i % 4])
    torch.cuda.synchronize()
    thr = (time.perf_counter() - t0) / 20
    print('batch %3d: host enqueue %.2f ms (min %.2f)  enqueue+drain %.2f ms  pipelined %.2f ms/step' %
          (bs, 1e3 * np.median(enq), 1e3 * min(enq), 1e3 * np.median(tot), 1e3 * thr), flush=True)
