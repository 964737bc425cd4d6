"""cProfile of the host side of one train step (batch 8: where the host is closest to being the limiter)."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnr_amd.config import make_config
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer
bs = 8
cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=%d' % bs], corpus_sizes=dict(vocabulary_size=60000))
torch.manual_seed(1)
model = Model(cfg, torch.randn(cfg.vocabulary_size, cfg.word_embedding_dim) * 0.3); model.initialize(); model = model.cuda().train()
tr = Trainer(model, cfg)
corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size))
rng = np.random.default_rng(3)
bts = [to_torch(corpus.batch(bs, rng), 'cuda') for _ in range(4)]
for i in range(6):
    tr.train_step(bts[i % 4])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    tr.train_step(bts[i % 4])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
