#!/usr/bin/env python3
"""Several data-parallel ranks of the PRODUCT path against the CPU oracle (test infrastructure; launched by the tests below with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 --master-port P tests/dp_rank_main.py --backend B)

  --backend nccl : one rank per GPU, RCCL over xGMI -- the configuration of a real data-parallel job (reference: trainer.py:212-220,
                   297).  Needs >= W GPUs (tests/test_hip_dp_gpu.py runs it when torch.cuda.device_count() >= 2).  With
                   NNR_DP_NATIVE=1 the exchange goes through the C-ABI's nnr_dp_* instead of torch.distributed.
  --backend gloo : the ranks SHARE GPU 0 and exchange through gloo (RCCL refuses two ranks on one device): everything but RCCL
                   itself -- rank set-up, sharding, the bucketed GradientExchange (early / table / late), clip+Adam's 1/world -- on
                   the 1-GPU lease.

Checks (rank 0 prints one JSON line; exit code 1 on failure):
  A. gradients: the exchanged gradient x 1/world of a CNE+SUE step at full model dimensions == the MEAN OVER RANKS OF THE ORACLE's
     per-shard gradients (every rank runs the CPU oracle on its own shard; CNE sorts ITS OWN shard, so the reference quantity is
     per shard, exactly as under the reference's DDP), and == the mean of the product's own exchange-free shard gradients;
  B. parameters: ranks initialised DIFFERENTLY are bit-identical after the constructor's broadcast and stay so over two steps;
  C. loader: two epochs over a reference-built tiny corpus (tests/golden/corpus_tiny_h50_sym.npz) with the per-rank loader
     semantics of trainer.py:252-258 -- every rank draws the SAME negative samples (identical numpy seed, MIND_dataset.py:27-47),
     DistributedSampler.set_epoch(e) order (dp.sampler_indices), per-rank batch = batch_size // world, last partial batch kept --
     DeviceCorpus.train_batch -> Trainer.train_step on every rank, against the oracle stepping with the mean of the ranks' oracle
     gradients: per-step loss of every rank and the final parameters;
  D. MHSA+MHSA (BASELINE configs[1]) through the NATIVE step at a per-rank batch where the user encoder's weight-gradient GEMMs run on
     the leaf stream (B * 50 * 32 >= ops.LEAF_MIN_ROWS): the exchanged gradient == the mean of the ranks' exchange-free shard
     gradients, bit-identical on every rank, and the early bucket's all-reduce is ordered behind the leaf stream (round-4 advisor,
     high: it was issued behind the main stream only)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

from nnr_amd import dp, ops
from nnr_amd.config import make_config
from nnr_amd.model import Model, negative_log_softmax
from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
from nnr_amd.trainer import Trainer
from oracle import corpus_oracle as CO, nnr_oracle as O           # checker only

ap = argparse.ArgumentParser()
ap.add_argument('--backend', default='gloo', choices=['gloo', 'nccl'])
ap.add_argument('--skip_epoch', action='store_true')
ap.add_argument('--only_epoch', action='store_true', help='diagnostics: part C only')
ap.add_argument('--skip_mhsa', action='store_true', help='skip part D (MHSA+MHSA native step)')
ap.add_argument('--trace2', action='store_true', help='diagnostics: device-side per-step checksums of the parameter buckets, compared at the end (no extra host sync)')
ap.add_argument('--trace', action='store_true', help='diagnostics: per step, which bucket of the parameters differs between the ranks')
args = ap.parse_args()

rank, local, world = dp.init_from_env(args.backend)
if args.backend == 'gloo':
    torch.cuda.set_device(0)                                     # the ranks share the lease's GPU
host = dist.new_group(backend='gloo') if args.backend == 'nccl' else None      # host-side exchange of the oracle's results
dev = torch.device('cuda', torch.cuda.current_device())
O.BiLSTM.backend = 'aten'
torch.set_num_threads(max(1, min(16, (os.cpu_count() or 8) // world)))      # (more threads are SLOWER on the oracle's mid-size ops)


def gather_mean(arrays):
    """Mean over the ranks of a list of numpy arrays (host side)."""
    box = [None] * world
    dist.all_gather_object(box, arrays, group=host)
    return [sum(b[i].astype(np.float64) for b in box) / world for i in range(len(arrays))]


def all_equal(obj):
    box = [None] * world
    dist.all_gather_object(box, obj, group=host)
    return all(np.array_equal(box[0], b) for b in box[1:])


# --------------------------------------------------------------------------------------------- A / B: full model dimensions
B = 4 * world
cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=%d' % B, '--world_size=%d' % world],
                  corpus_sizes=dict(vocabulary_size=3000), dropout_rate=0.0, tie_order='stable')
full_np = SynthCorpus(SynthSpec(vocabulary_size=3000, news_pool=1500)).batch(B, np.random.default_rng(5))
full = to_torch(full_np, dev)
shard = dp.shard_batch(full, rank, world)
shard_cpu = dp.shard_batch(to_torch(full_np), rank, world)


def build(seed):
    torch.manual_seed(seed)
    m = Model(cfg)
    m.initialize()
    with torch.no_grad():
        for p in m.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0, 0.05)
    return m


def backward_only(tr, batch):
    tr.exchange.begin_step()
    tr.flat.zero_grad()
    loss = negative_log_softmax(tr.model(*[t.clone() for t in batch]))
    loss.backward()
    ops.join_extra_streams()


err_prod = err_oracle = err_touched = 0.0
touched_stats = None
worst_name, params_same, rccl_ranks = '', True, None
if not args.only_epoch:
    cpu_model = build(0)
    ref = O.Model(cfg)
    ref.load_state_dict(cpu_model.state_dict())
    ref.train()
    rl = O.negative_log_softmax(ref(*[t.clone() for t in shard_cpu]))
    rl.backward()
    names = [k for k, _ in ref.named_parameters()]
    want_oracle = dict(zip(names, gather_mean([p.grad.numpy() for _, p in ref.named_parameters()])))

    noex = Trainer(build(0).to(dev).train(), cfg)                  # exchange-free gradient of this rank's shard, averaged by hand
    noex.exchange.active = lambda: False
    backward_only(noex, shard)
    want_prod = noex.flat.grad.clone()
    dist.all_reduce(want_prod)
    want_prod /= world

    tr0 = Trainer(build(0).to(dev).train(), cfg)
    assert tr0.exchange.active() and tr0.exchange.early_span is not None and tr0.exchange.table_span is not None
    backward_only(tr0, shard)
    scale = tr0.exchange.finish()
    got = tr0.flat.grad * scale
    torch.cuda.synchronize()
    err_prod = float((got - want_prod).abs().max()) / max(1e-12, float(want_prod.abs().max()))
    total = float(np.sqrt(sum(float((g ** 2).sum()) for g in want_oracle.values())))
    err_oracle, worst_name = 0.0, ''
    for k, p in tr0.model.named_parameters():
        if k.startswith('user_encoder.news_encoder.'):
            continue
        g = (p.grad.detach() * scale).cpu().double().numpy()
        w = want_oracle[k]
        e = float(np.abs(g - w).max()) / max(1e-3, 0.05 * total, float(np.linalg.norm(w)))
        if e > err_oracle:
            err_oracle, worst_name = e, k

    # the same exchange with the table bucket as a TOUCHED-ROW exchange (NNR_DP_TOUCHED_ROWS=1: flags summed over the ranks, the union's
    # rows packed, all-reduced, written back): the dense gradient every rank ends up with must be the same mean
    touched_stats, err_touched = None, 0.0
    os.environ['NNR_DP_TOUCHED_ROWS'] = '1'
    try:
        tr1 = Trainer(build(0).to(dev).train(), cfg)
        assert tr1.exchange.touched
        backward_only(tr1, shard)
        scale1 = tr1.exchange.finish()
        torch.cuda.synchronize()
        err_touched = float((tr1.flat.grad * scale1 - want_prod).abs().max()) / max(1e-12, float(want_prod.abs().max()))
        touched_stats = tr1.exchange.describe().get('touched_rows_last_step')
    finally:
        del os.environ['NNR_DP_TOUCHED_ROWS']

    tr = Trainer(build(100 + rank).to(dev).train(), cfg)         # different initial parameters per rank: the constructor's broadcast fixes that
    for _ in range(2):
        tr.train_step([t.clone() for t in shard])
    torch.cuda.synchronize()
    params_same = all_equal(tr.flat.flat.cpu().numpy()) and bool(torch.isfinite(tr.flat.flat).all())
    rccl_ranks = None
    if args.backend == 'nccl':
        rccl_ranks = dist.get_world_size()
        assert dist.get_backend() == 'nccl'

# --------------------------------------------------------------------------------------------- D: MHSA+MHSA native step, leaf-deferred weight gradients
mhsa = None
if not args.skip_mhsa and not args.only_epoch:
    from nnr_amd import step as native_step
    Bm = 32 * world                                              # per-rank 32: 32 * 50 * 32 = 51 200 token rows >= ops.LEAF_MIN_ROWS -> deferral on
    mcfg = make_config(['--news_encoder=MHSA', '--user_encoder=MHSA', '--dataset=200k', '--batch_size=%d' % Bm, '--world_size=%d' % world],
                       corpus_sizes=dict(vocabulary_size=3000), dropout_rate=0.2)
    mfull = to_torch(SynthCorpus(SynthSpec(vocabulary_size=3000, news_pool=1500)).batch(Bm, np.random.default_rng(11)), dev)
    mshard = dp.shard_batch(mfull, rank, world)

    def mbuild():
        torch.manual_seed(0)
        m = Model(mcfg)
        m.initialize()
        return m.to(dev).train()

    def native_grad(tr):
        assert native_step.kind(tr.model) == 'mhsa'
        tr.exchange.begin_step()
        tr._zero_grad_aside()
        before = ops._DEFER['calls']
        native_step.forward_backward(tr, [t.clone() for t in mshard])
        return tr.exchange.finish(), ops._DEFER['calls'] - before

    t_no = Trainer(mbuild(), mcfg)
    t_no.exchange.active = lambda: False
    native_grad(t_no)
    want = t_no.flat.grad.clone()
    dist.all_reduce(want)
    want /= world
    worst, deferred = 0.0, 0
    same = True
    for rep in range(3):                                         # (the race was timing-dependent: a few repetitions)
        t_ex = Trainer(mbuild(), mcfg)
        assert t_ex.exchange.active()
        # (in today's flat layout the MHSA user encoder's parameters are NOT contiguous -- the fused W_Q|W_K|W_V groups of both encoders go
        # first --, so early_span is None and everything is reduced by finish() behind the step's final join; the leaf-stream join in front
        # of the hook is what keeps the step correct if a layout change ever makes the early bucket exist: reported below)
        sc, deferred = native_grad(t_ex)
        got = t_ex.flat.grad * sc
        torch.cuda.synchronize()
        worst = max(worst, float((got - want).abs().max()) / max(1e-12, float(want.abs().max())))
        same &= all_equal(t_ex.flat.grad.cpu().numpy())
    mhsa = {'per_rank_batch': Bm // world, 'early_bucket': t_ex.exchange.early_span is not None, 'leaf_deferred_launches': deferred, 'grad_rel_err_vs_mean_of_shard_gradients': worst,
            'gradients_identical_across_ranks': bool(same), 'ok': bool(worst <= 2e-5 and same and deferred > 0)}

# --------------------------------------------------------------------------------------------- C: two epochs over a tiny corpus
epoch = None
if not args.skip_epoch:
    from nnr_amd.corpus import DeviceCorpus, negative_sampling
    c = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'corpus_tiny_h50_sym.npz')))
    V = int(max(c['news_title_text'].max(), c['news_abstract_text'].max())) + 1
    gb = 2 * world                                               # --batch_size (global); per rank gb // world (trainer.py:218)
    tcfg = O.default_config(news_encoder='CNE', user_encoder='SUE', dataset='small', vocabulary_size=V, word_embedding_dim=16, hidden_dim=8,
                            attention_dim=8, max_history_num=int(c['max_history_num']), max_title_length=8, max_abstract_length=16,
                            category_num=int(c['category_num']), subCategory_num=int(c['news_subCategory'].max()) + 1, category_embedding_dim=4,
                            subCategory_embedding_dim=4, negative_sample_num=4, head_num=2, head_dim=4, cnn_kernel_num=12, gcn_layer_num=2,
                            dropout_rate=0.0, lr=1e-2, user_num=int(c['beh_user'].max()) + 1, tie_order='stable', batch_size=gb, world_size=world)
    torch.manual_seed(3)
    oref = O.Model(tcfg)
    oref.initialize()
    with torch.no_grad():
        for p in oref.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0, 0.05)
            p.mul_(1.5)
    oref.train()
    model = Model(tcfg, torch.zeros(V, 16))
    model.load_state_dict(oref.state_dict())
    trainer = Trainer(model.to(dev).train(), tcfg)
    opt = O.make_optimizer(oref, tcfg)
    dc = DeviceCorpus(c, dev, int(c['category_num']), graph='build', norm='symmetric')
    n = int(c['beh_user'].shape[0])
    news = int(c['news_category'].shape[0])
    lists = np.random.default_rng(7)                            # (click, non-clicked news) per behaviour: 1..9 negatives -> both sampler branches
    beh = [(int(c['train_samples'][i, 0]), lists.integers(1, news, size=int(lists.integers(1, 10))).tolist()) for i in range(n)]
    rs = np.random.RandomState(int(tcfg.seed))                  # every rank seeds numpy identically (config.py:125-128)
    per = gb // world
    worst_loss, steps, samples_same, covered = 0.0, 0, True, True
    for e in range(2):
        samples = negative_sampling(beh, 4, rs.randint)         # trainer.py:255: every rank re-samples the whole dataset
        samples_same &= all_equal(samples)
        dc.set_samples(samples)
        order = dp.sampler_indices(n, rank, world, epoch=e, seed=0).numpy()      # DistributedSampler(...).set_epoch(e), trainer.py:256-257
        box = [None] * world
        dist.all_gather_object(box, order.tolist(), group=host)
        covered &= sorted(set(sum(box, []))) == list(range(n)) and len(sum(box, [])) == -(-n // world) * world
        cc = dict(c, train_samples=samples)
        for s in range(0, len(order), per):
            idx = order[s:s + per].astype(np.int32)
            _, loss = trainer.train_step(dc.train_batch(idx))
            rb = [torch.from_numpy(np.ascontiguousarray(a)) for a in CO.train_batch(cc, idx)]
            rloss = O.negative_log_softmax(oref(*rb))
            opt.zero_grad()
            rloss.backward()
            mean = gather_mean([p.grad.numpy() for p in oref.parameters()])      # DDP's gradient averaging, on the host
            for p, g in zip(oref.parameters(), mean):
                p.grad = torch.from_numpy(g.astype(np.float32))
            torch.nn.utils.clip_grad_norm_(oref.parameters(), tcfg.gradient_clip_norm)
            opt.step()
            worst_loss = max(worst_loss, abs(float(loss) - float(rloss)))
            steps += 1
            if args.trace2:
                ex = trainer.exchange
                regions = [('early', ex.early_span), ('table', ex.table_span)] + [('late%d' % i, sp) for i, sp in enumerate(ex.late_spans)]
                if steps == 1:
                    chk = torch.zeros((64, len(regions), 2), device=dev, dtype=torch.float64)
                    chk_paths = []
                for ri, (_, (a_, b_)) in enumerate(regions):
                    chk[steps - 1, ri, 0] = trainer.flat.flat[a_:b_].double().abs().sum()          # enqueued behind the step, not read here
                    chk[steps - 1, ri, 1] = trainer.flat.grad[a_:b_].double().abs().sum()
                chk_paths.append(trainer.last_path)
            if args.trace:
                ex = trainer.exchange
                regions = [('early', ex.early_span), ('table', ex.table_span)] + [('late%d' % i, sp) for i, sp in enumerate(ex.late_spans)]
                sums = [float(trainer.flat.flat[a:b].double().abs().sum()) for _, (a, b) in regions]
                gsum = [float(trainer.flat.grad[a:b].double().abs().sum()) for _, (a, b) in regions]
                box2 = [None] * world
                dist.all_gather_object(box2, (sums, gsum, trainer.last_path, float(loss), float(rloss)), group=host)
                if rank == 0:
                    diff = [n for (n, _), x, y in zip(regions, box2[0][0], box2[1][0]) if x != y]
                    gdiff = [n for (n, _), x, y in zip(regions, box2[0][1], box2[1][1]) if x != y]
                    print('trace epoch %d step %d paths %s/%s loss %.6f/%.6f oracle %.6f/%.6f params differ in %s, exchanged grads differ in %s' % (
                        e, steps, box2[0][2], box2[1][2], box2[0][3], box2[1][3], box2[0][4], box2[1][4], diff, gdiff), flush=True)
    if args.trace2:
        box3 = [None] * world
        dist.all_gather_object(box3, (chk[:steps].cpu().numpy(), chk_paths), group=host)
        if rank == 0:
            for st_ in range(steps):
                pd = [n for ri, (n, _) in enumerate(regions) if box3[0][0][st_, ri, 0] != box3[1][0][st_, ri, 0]]
                gd = [n for ri, (n, _) in enumerate(regions) if box3[0][0][st_, ri, 1] != box3[1][0][st_, ri, 1]]
                print('trace2 step %d paths %s/%s params differ in %s, exchanged grads differ in %s' % (st_ + 1, box3[0][1][st_], box3[1][1][st_], pd, gd), flush=True)
    rp = dict(oref.named_parameters())
    worst_param = max(float((p.detach().cpu() - rp[k].detach()).abs().max()) for k, p in trainer.model.named_parameters()
                      if not k.startswith('user_encoder.news_encoder.'))
    box = [None] * world
    dist.all_gather_object(box, (worst_loss, worst_param), group=host)
    epoch = {'steps_per_rank': steps, 'negative_samples_identical_across_ranks': bool(samples_same), 'sampler_covers_every_behaviour': bool(covered),
             'worst_loss_diff_vs_oracle': max(b[0] for b in box), 'worst_param_diff_vs_oracle': max(b[1] for b in box),
             'parameters_identical_across_ranks': all_equal(trainer.flat.flat.cpu().numpy()),
             'ok': bool(samples_same and covered and max(b[0] for b in box) <= 5e-5 and max(b[1] for b in box) <= steps * 1e-2 * 1.01 + 1e-4)}
    epoch['ok'] = epoch['ok'] and epoch['parameters_identical_across_ranks']

tmo = ops.lstm_sync_timeouts()
ok = err_prod <= 2e-5 and err_touched <= 2e-5 and err_oracle <= 1e-4 and params_same and (epoch is None or epoch['ok']) and (mhsa is None or mhsa['ok']) and tmo == 0
if rank == 0:
    print(json.dumps({'world': world, 'backend': args.backend, 'rccl_ranks': rccl_ranks, 'devices': torch.cuda.device_count(),
                      'binding': (trainer if args.only_epoch else tr).exchange.describe()['binding'], 'buckets': [b['name'] for b in (trainer if args.only_epoch else tr).exchange.describe()['buckets']],
                      'grad_rel_err_vs_mean_of_shard_gradients': err_prod, 'grad_err_vs_oracle_mean_of_shard_gradients': err_oracle,
                      'touched_rows': touched_stats, 'grad_rel_err_touched_row_exchange': err_touched,
                      'worst_gradient': worst_name, 'recurrence_exchange_timeouts': tmo, 'parameters_identical_across_ranks': params_same, 'mhsa_native': mhsa, 'epoch': epoch, 'ok': bool(ok)}))
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
