import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnr_amd import profile as prof
from nnr_amd.config import make_config
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer
dev = torch.device('cuda')
cfg = make_config([], corpus_sizes=dict(vocabulary_size=60000))
torch.manual_seed(0)
table = torch.randn(cfg.vocabulary_size, 300) * 0.3
model = Model(cfg, table); model.initialize(); model = model.to(dev).train()
tr = Trainer(model, cfg)
corpus = SynthCorpus(SynthSpec(vocabulary_size=60000))
rng = np.random.default_rng(100)
bs = [to_torch(corpus.batch(64, rng), dev) for _ in range(4)]
for i in range(3): tr.train_step(bs[i % 4])
torch.cuda.synchronize(); prof.enable()
steps = 5
for i in range(steps): tr.train_step(bs[i % 4])
prof.disable()
rows = sorted(prof.by_shape().items(), key=lambda kv: -kv[1]['ms'])
tot = sum(v['ms'] for _, v in rows)
print('instrumented ms/step %.2f' % (tot / steps))
for (fam, tag), v in rows[:40]:
    print('%-16s %-34s %3d calls/step %7.3f ms/step %6.1f TF' % (fam, tag, v['launches'] // steps, v['ms'] / steps, v['flops'] / (v['ms'] * 1e-3) / 1e12 if v['ms'] else 0))
