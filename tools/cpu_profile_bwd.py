"""Host time of the backward pieces (they run on autograd's worker thread, invisible to a main-thread cProfile)."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnr_amd import news_encoders as NE, user_encoders as UE
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
acc = {}
pr = cProfile.Profile()
def wrap(mod, name, prof=False):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter()
        if prof: pr.enable()
        r = f(*a, **k)
        if prof: pr.disable()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    setattr(mod, name, g)
wrap(NE, 'cne_backward_many', prof=True); wrap(UE, 'sue_backward'); wrap(NE, 'cne_forward_many'); wrap(UE, 'sue_forward')
for i in range(6):
    tr.train_step(bts[i % 4])
torch.cuda.synchronize(); acc.clear(); pr.clear()
n = 20
t0 = time.perf_counter()
for i in range(n):
    tr.train_step(bts[i % 4])
tot = time.perf_counter() - t0
torch.cuda.synchronize()
print('host time per step %.2f ms: ' % (tot / n * 1e3) + ', '.join('%s %.2f' % (k, v / n * 1e3) for k, v in acc.items()))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:3000])
