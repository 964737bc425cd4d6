#!/usr/bin/env python3
"""Single-GPU ordering check of the bucketed gradient exchange (dp.GradientExchange) on a ONE-rank RCCL communicator
(NNR_DP_FORCE=1): the early bucket (the user encoder's gradients) must be handed to RCCL while the news-encoder backward is still
ahead, and the step must give the same parameters as the exchange-free step.  Prints one JSON line.
Usage: dp_overlap_check.py [torch|native]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', '29533'), NNR_DP_FORCE='1')
binding = sys.argv[1] if len(sys.argv) > 1 else 'torch'
if binding == 'native':
    os.environ['NNR_DP_NATIVE'] = '1'
import numpy as np, torch
import torch.distributed as dist
from nnr_amd.config import make_config
from nnr_amd.model import Model
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer

torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='env://', world_size=1, rank=0)
B = 32
cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=%d' % B], corpus_sizes=dict(vocabulary_size=5000),
                  dropout_rate=0.0, tie_order='stable')
batch = to_torch(SynthCorpus(SynthSpec(vocabulary_size=5000, news_pool=2000)).batch(B, np.random.default_rng(1)), 'cuda')


def build():
    torch.manual_seed(0)
    m = Model(cfg); m.initialize()
    return m.cuda().train()


tr = Trainer(build(), cfg)
assert tr.exchange.active() and tr.exchange.early_span is not None
tr.train_step(batch)
g_first = tr.flat.grad.clone()                 # gradients of the first step (after the one-rank all-reduce = identity)
tr.train_step(batch)
tr.exchange.events = {}
start = torch.cuda.Event(enable_timing=True); start.record()
tr.train_step(batch)
torch.cuda.synchronize()
ev = tr.exchange.events
t_early, t_done = start.elapsed_time(ev['early_issued']), start.elapsed_time(ev['finished'])
t_table = start.elapsed_time(ev['table_issued']) if 'table_issued' in ev else None
os.environ['NNR_DP_FORCE'] = '0'
ref = Trainer(build(), cfg)
ref.exchange.force = False
assert not ref.exchange.active()
ref.train_step(batch)
torch.cuda.synchronize()
# same initial parameters, same batch: the first step's gradients must agree (f32 atomics reorder sums: relative 1e-5 of the norm)
diff = float((g_first - ref.flat.grad).abs().max()) / max(1e-12, float(ref.flat.grad.norm()))
print(json.dumps({'binding': binding, 'early_bucket_issued_ms': round(t_early, 3), 'exchange_finished_ms': round(t_done, 3),
                  'backward_left_when_early_bucket_went_out_ms': round(t_done - t_early, 3),
                  'table_bucket_issued_ms': None if t_table is None else round(t_table, 3),
                  'step_left_when_table_bucket_went_out_ms': None if t_table is None else round(t_done - t_table, 3), 'buckets': tr.exchange.describe(),
                  'max_grad_diff_vs_no_exchange_rel': diff}))
dist.destroy_process_group()
