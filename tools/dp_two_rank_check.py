#!/usr/bin/env python3
"""Two data-parallel ranks of the PRODUCT path on ONE GPU (test mode: RCCL refuses two ranks on a device, so the exchange goes through
gloo; everything else -- rank set-up, sharding, the bucketed GradientExchange with its early / table / late slices, clip+Adam with the 1/world
scale -- is the code a multi-GPU job runs).  Launch:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/dp_two_rank_check.py
Checks (rank 0 prints one JSON line, exit code 1 on failure):
  * the exchanged gradient x 1/world == the mean over ranks of the exchange-free per-shard gradients (CNE sorts ITS OWN shard, so the
    reference quantity is per shard, exactly as under the reference's DDP);
  * after two optimizer steps the parameters of the two ranks are bit-identical."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from nnr_amd import dp, ops
from nnr_amd.config import make_config
from nnr_amd.model import Model, negative_log_softmax
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer

rank, local, world = dp.init_from_env('gloo')
torch.cuda.set_device(0)
B = 8
cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=%d' % B, '--world_size=%d' % world],
                  corpus_sizes=dict(vocabulary_size=3000), dropout_rate=0.0, tie_order='stable')
full = to_torch(SynthCorpus(SynthSpec(vocabulary_size=3000, news_pool=1500)).batch(B, np.random.default_rng(5)), 'cuda')
shard = dp.shard_batch(full, rank, world)


def build(seed):
    torch.manual_seed(seed)
    m = Model(cfg); m.initialize()
    return m.cuda().train()


def backward_only(tr, batch):
    tr.flat.zero_grad()
    loss = negative_log_softmax(tr.model(*[t.clone() for t in batch]))
    loss.backward()
    ops.join_extra_streams()


# exchange-free gradient of this rank's shard, averaged over the ranks by hand
ref = Trainer(build(0), cfg)
ref.exchange.active = lambda: False
backward_only(ref, shard)
want = ref.flat.grad.clone()
dist.all_reduce(want)
want /= world
# the product's exchange
tr = Trainer(build(100 + rank), cfg)          # different initial parameters per rank: the broadcast in Trainer.__init__ must fix that
assert tr.exchange.active() and tr.exchange.early_span is not None and tr.exchange.table_span is not None
p_ref = ref.flat.flat.clone()
dist.broadcast(p_ref, src=0)
tr0 = Trainer(build(0), cfg)                  # same parameters as `ref` on every rank
backward_only(tr0, shard)
scale = tr0.exchange.finish()
got = tr0.flat.grad * scale
torch.cuda.synchronize()
err = float((got - want).abs().max()) / max(1e-12, float(want.abs().max()))
# two full steps on the rank-dependent model: parameters must stay identical across ranks
for _ in range(2):
    tr.train_step([t.clone() for t in shard])
p = tr.flat.flat.clone()
p0 = p.clone()
dist.broadcast(p0, src=0)
same = torch.tensor([float(torch.equal(p, p0))], device='cuda')
dist.all_reduce(same, op=dist.ReduceOp.MIN)
ok = err <= 2e-5 and bool(same.item()) and bool(torch.isfinite(p).all())
if rank == 0:
    print(json.dumps({'world': world, 'buckets': [b['name'] for b in tr.exchange.describe()['buckets']], 'grad_rel_err_vs_mean_of_shard_gradients': err,
                      'parameters_identical_across_ranks': bool(same.item()), 'ok': ok}))
dist.destroy_process_group()
sys.exit(0 if ok else 1)
