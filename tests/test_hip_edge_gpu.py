"""Edge cases and bench-shaped batches at FULL model dimensions, HIP path vs the CPU oracle (the reference has no tests of
its own; these are the ragged / empty / maximum-size inputs its data pipeline can produce)."""
import numpy as np
import pytest
import torch

from nnr_amd.config import make_config
from nnr_amd.synth import SynthSpec, SynthCorpus, BATCH_FIELDS, to_torch

pytestmark = pytest.mark.gpu


def _models(cfg, seed=0):
    from nnr_amd.model import Model
    from oracle import nnr_oracle as O
    torch.manual_seed(seed)
    ref = O.Model(cfg)
    ref.initialize()
    with torch.no_grad():
        for p in ref.parameters():                 # zero-initialised tensors (proxy nodes, biases) get signal too
            if float(p.abs().max()) == 0.0:
                p.normal_(0, 0.05)
    ref.train()
    model = Model(cfg)
    model.load_state_dict(ref.state_dict())
    return model.cuda().train(), ref


def _cfg(**kw):
    return make_config(['--news_encoder=CNE', '--user_encoder=SUE'], corpus_sizes=dict(vocabulary_size=800), dropout_rate=0.0,
                       tie_order='stable', **kw)


def _edge_batch(cfg):
    """sample 0: empty history; sample 1: 50 history news, every title / abstract at maximum length; sample 2: every sequence of
    length 1 and all candidates identical; sample 3: ordinary."""
    spec = SynthSpec(vocabulary_size=cfg.vocabulary_size, news_pool=300, seed=5)
    normal = SynthCorpus(spec)
    dense = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size, news_pool=120, seed=6, dense=True))
    rng = np.random.default_rng(3)
    b = normal.batch(4, rng)
    bd = dense.batch(1, rng)
    H = spec.max_history_num
    # sample 1 <- dense corpus, full history
    hist = rng.integers(1, dense.spec.news_pool, size=H)
    g, cm, ci = dense.history_graph(dense.category[hist], H)
    for k in BATCH_FIELDS:
        if k.startswith('news_'):
            b[k][1] = bd[k][0]
    b['user_title_text'][1], b['user_title_mask'][1] = dense.title_text[hist], dense.title_mask[hist]
    b['user_content_text'][1], b['user_content_mask'][1] = dense.content_text[hist], dense.content_mask[hist]
    b['user_category'][1], b['user_subCategory'][1] = dense.category[hist], dense.subCategory[hist]
    b['user_history_mask'][1] = True
    b['user_history_graph'][1], b['user_history_category_mask'][1], b['user_history_category_indices'][1] = g, cm, ci
    # sample 0 <- empty history (all <PAD> news, identity graph)
    g, cm, ci = normal.history_graph(np.zeros(H, np.int32), 0)
    for k in ('user_title_text', 'user_content_text', 'user_category', 'user_subCategory'):
        b[k][0] = 0
    b['user_title_mask'][0] = False
    b['user_content_mask'][0] = False
    b['user_title_mask'][0][:, 0] = True
    b['user_content_mask'][0][:, 0] = True
    b['user_history_mask'][0] = False
    b['user_history_graph'][0], b['user_history_category_mask'][0], b['user_history_category_indices'][0] = g, cm, ci
    # sample 2 <- length-1 sequences everywhere, identical candidates
    for k in ('user_title_mask', 'user_content_mask', 'news_title_mask', 'news_content_mask'):
        b[k][2][:, 1:] = False
    for k in ('user_title_text', 'user_content_text', 'news_title_text', 'news_content_text'):
        b[k][2][:, 1:] = 0
    for k in BATCH_FIELDS:
        if k.startswith('news_'):
            b[k][2][:] = b[k][2][0]
    return {k: np.ascontiguousarray(v) for k, v in b.items()}


def _compare(model, ref, batch, tol=1e-4):
    from nnr_amd.model import negative_log_softmax
    from oracle import nnr_oracle as O
    for p in model.parameters():
        p.grad = None
    logits = model(*to_torch(batch, 'cuda'))
    loss = negative_log_softmax(logits)
    loss.backward()
    rl = ref(*to_torch(batch))
    rloss = O.negative_log_softmax(rl)
    ref.zero_grad()
    rloss.backward()
    err = float((logits.detach().cpu() - rl.detach()).abs().max())
    assert err <= tol, 'logits differ by %.3e' % err
    assert abs(float(loss) - float(rloss)) <= tol
    total = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in ref.parameters())))
    rp = dict(ref.named_parameters())
    for k, p in model.named_parameters():
        g, rg = p.grad.detach().cpu().double(), rp[k].grad.double()
        assert float((g - rg).abs().max()) <= 1e-4 * max(1e-3, 0.05 * total, float(rg.norm())), 'grad ' + k
    return logits.detach()


def test_edge_cases_empty_max_and_unit_lengths():
    cfg = _cfg(batch_size=4)
    model, ref = _models(cfg)
    _compare(model, ref, _edge_batch(cfg))


def test_bench_shaped_batch_matches_oracle():
    cfg = _cfg(batch_size=8)
    model, ref = _models(cfg, seed=1)
    corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size, news_pool=2000, seed=9))
    _compare(model, ref, corpus.batch(8, np.random.default_rng(4)))


@pytest.mark.parametrize('hidden,ln', [(48, False), (128, False), (224, False), (256, True)])
def test_hidden_dims_with_full_width_word_rows(hidden, ln):
    """--hidden_dim values whose buffers / tiles differ from the default 200 at the reference's word_embedding_dim 300 and a token
    capacity >= 8 192 rows (B = 2: 110 news x 128): hidden <= 144 makes the cell-state buffer [cap, 2*HP] SMALLER than the
    embedding-row gradient [cap, 300] staged in it (round-2 advisor finding: news_encoders.py dx_scatter), hidden 212..256 puts
    the gathered dW_hh GEMM (N = hidden) past the 208-column tile (ops.tn_tile fell through to a tile without a gather path);
    hidden 256 makes the news vector 1 124 wide (pool / LayerNorm kernels were limited to 1 024 columns)."""
    cfg = _cfg(batch_size=2, hidden_dim=hidden, gcn_layer_norm=ln)
    assert cfg.word_embedding_dim == 300
    model, ref = _models(cfg, seed=hidden)
    corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size, news_pool=400, seed=12))
    _compare(model, ref, corpus.batch(2, np.random.default_rng(hidden)))


def test_forward_bitwise_deterministic_and_backward_accumulates():
    from nnr_amd.model import negative_log_softmax
    cfg = _cfg(batch_size=4)
    model, _ = _models(cfg, seed=2)
    batch = _edge_batch(cfg)
    a = model(*to_torch(batch, 'cuda')).detach()
    b = model(*to_torch(batch, 'cuda')).detach()
    assert torch.equal(a, b), 'forward must be bitwise reproducible (no atomics on the forward path)'
    for p in model.parameters():
        p.grad = None
    negative_log_softmax(model(*to_torch(batch, 'cuda'))).backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    negative_log_softmax(model(*to_torch(batch, 'cuda'))).backward()          # second backward ACCUMULATES like autograd
    for k, p in model.named_parameters():
        scale = max(1e-6, float(g1[k].abs().max()))
        assert float((p.grad - 2 * g1[k]).abs().max()) <= 1e-5 * scale, k


def test_mhsa_and_cnn_pairs_on_ragged_batch():
    """MHSA+MHSA and CNN+ATT at full dims on a MIND-shaped batch incl. an empty history (fully masked attention rows)."""
    from nnr_amd.model import Model, negative_log_softmax
    from oracle import nnr_oracle as O
    for ne, ue in (('MHSA', 'MHSA'), ('CNN', 'ATT')):
        cfg = make_config(['--news_encoder=' + ne, '--user_encoder=' + ue], corpus_sizes=dict(vocabulary_size=800), dropout_rate=0.0)
        torch.manual_seed(3)
        ref = O.Model(cfg)
        ref.initialize()
        ref.eval()                                  # MHSA-user has a hard-wired F.dropout(p=0.5) in train mode
        model = Model(cfg)
        model.load_state_dict(ref.state_dict())
        model = model.cuda().eval()
        batch = _edge_batch(cfg)
        logits = model(*to_torch(batch, 'cuda'))
        rl = ref(*to_torch(batch))
        assert float((logits.detach().cpu() - rl.detach()).abs().max()) <= 1e-4, (ne, ue)
        negative_log_softmax(logits).backward()
        O.negative_log_softmax(rl).backward()
        rp = dict(ref.named_parameters())
        total = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in ref.parameters())))
        for k, p in model.named_parameters():
            assert float((p.grad.cpu().double() - rp[k].grad.double()).abs().max()) <= 1e-4 * max(1e-3, 0.05 * total), (ne, ue, k)


@pytest.mark.parametrize('dropout', [0.0, 0.2])
def test_mhsa_pair_with_fully_masked_and_gapped_titles(dropout):
    """The MHSA news encoder runs over packed token rows (round 5); padding is observable in exactly one case -- a title whose mask is
    ALL zero (no mask[:, 0] = 1 fix-up exists in the MHSA encoder, newsEncoders.py:187-200: every key is -1e9, both softmaxes are uniform
    over all 32 positions).  The corpus never produces one (MIND_corpus.py:352 gives the <PAD> news one valid position), the kernels must
    still agree with the reference's semantics: a fully masked history title, a fully masked CANDIDATE title, and a title with an
    interior masked position, forward + every gradient against the oracle, with dropout off and on (HIP masks injected)."""
    import hip_masks
    cfg = make_config(['--news_encoder=MHSA', '--user_encoder=MHSA'], corpus_sizes=dict(vocabulary_size=800), dropout_rate=dropout, batch_size=4)
    model, ref = _models(cfg, seed=11)
    corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size, news_pool=400, seed=13))
    b = corpus.batch(4, np.random.default_rng(14))
    b['user_title_mask'][1, 7, :] = False                      # fully masked history title (its ids stay: the rows still exist)
    b['news_title_mask'][2, 3, :] = False                      # fully masked candidate title
    b['user_title_mask'][0, 2, :6] = True
    b['user_title_mask'][0, 2, 2] = False                      # an interior masked position
    b = {k: np.ascontiguousarray(v) for k, v in b.items()}
    # (train mode: the MHSA user encoder's hard-wired F.dropout(p = 0.5), userEncoders.py:171, is on in both parametrisations -- its mask,
    # and with dropout > 0 the news encoder's word / attention-output / category masks, come from the HIP generator)
    hip_masks.inject(model, ref, dict(zip(BATCH_FIELDS, to_torch(b, 'cuda'))))
    _compare(model, ref, b)


def test_deferred_weight_gradients_match_inline():
    """MHSA+MHSA and CNN+ATT on a GPU-bound step size (>= ops.LEAF_MIN_ROWS token rows): the weight-gradient GEMMs go to the
    leaf stream (ops.leaf_deferred), with W_Q|W_K|W_V fused through the trainer's flat layout, and the candidate encoder call
    runs on a side stream (model._JoinSideFn).  Same gradients as the inline, sequential, per-projection form on a model that
    was not re-homed (f32 atomics reorder sums: relative 1e-5).  Gradients are read on the stream that called backward():
    the end-of-pass callbacks must have joined every package stream into it."""
    from nnr_amd import ops
    from nnr_amd.model import Model, negative_log_softmax
    from nnr_amd.trainer import FlatParams
    for ne, ue in (('MHSA', 'MHSA'), ('CNN', 'ATT')):
        cfg = make_config(['--news_encoder=' + ne, '--user_encoder=' + ue], corpus_sizes=dict(vocabulary_size=800), dropout_rate=0.0)
        B = -(-ops.LEAF_MIN_ROWS // (cfg.max_history_num * cfg.max_title_length))
        corpus = SynthCorpus(SynthSpec(vocabulary_size=800, news_pool=300, seed=5))
        batch = corpus.batch(B, np.random.default_rng(8))
        grads = []
        for deferred in (False, True):
            torch.manual_seed(9)
            model = Model(cfg)
            model.initialize()
            model = model.cuda().eval()
            if deferred:
                FlatParams(model)                    # adjacent W_Q | W_K | W_V -> the fused projection path
            ops._DEFER['off'] = not deferred
            side_call, ops.SIDE_CALL = ops.SIDE_CALL, deferred      # ... and the candidate call on a side stream or in sequence
            calls = ops._DEFER['calls']
            try:
                negative_log_softmax(model(*to_torch(batch, 'cuda'))).backward()
            finally:
                ops._DEFER['off'], ops.SIDE_CALL = False, side_call
            assert (ops._DEFER['calls'] > calls) == deferred
            assert not ops._DEFER['keep'] and not ops._DEFER['queued']        # the end-of-pass callback ran
            grads.append({k: p.grad.clone() for k, p in model.named_parameters()})   # same stream as the join: ordered
        top = max(float(g.abs().max()) for g in grads[0].values())
        for k in grads[0]:                           # (W_K.bias has a mathematically zero gradient: rounding noise only)
            scale = max(1e-2 * top, float(grads[0][k].abs().max()))
            assert float((grads[0][k] - grads[1][k]).abs().max()) <= 2e-5 * scale, (ne, ue, k)


def test_large_vocabulary_config_parity_and_dropout_properties():
    """BASELINE.json config 5 (CNE+SUE 'large': vocabulary 130 000, per-GPU batch 16, dropout 0.1).  Dropout off: logits /
    loss / gradients against the oracle on a 2-impression shard (the oracle finishes in seconds).  Dropout on, full per-GPU
    batch: size-independent properties -- finite loss, the embedding gradient touches exactly the word rows of the batch,
    two runs with the same seed are identical in the forward pass, and the keep rate of the embedding dropout is 1 - p."""
    from nnr_amd import ops
    from nnr_amd.model import Model, negative_log_softmax
    from nnr_amd.trainer import Trainer
    V = 130000
    cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=large'], corpus_sizes=dict(vocabulary_size=V), dropout_rate=0.0,
                      tie_order='stable', batch_size=2)
    model, ref = _models(cfg, seed=4)
    corpus = SynthCorpus(SynthSpec(vocabulary_size=V, news_pool=600, seed=11))
    _compare(model, ref, corpus.batch(2, np.random.default_rng(6)))
    # dropout on (the reference's large-dataset setting), per-GPU batch of the 8-GPU configuration
    cfg2 = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=large', '--batch_size=128', '--world_size=8'],
                       corpus_sizes=dict(vocabulary_size=V))
    assert abs(cfg2.dropout_rate - 0.1) < 1e-9
    torch.manual_seed(0)
    m2 = Model(cfg2, torch.randn(V, cfg2.word_embedding_dim) * 0.3)
    m2.initialize()
    m2 = m2.cuda().train()
    batch = corpus.batch(16, np.random.default_rng(8))
    for p in m2.parameters():
        p.grad = None
    logits = m2(*to_torch(batch, 'cuda'))
    loss = negative_log_softmax(logits)
    loss.backward()
    assert bool(torch.isfinite(loss)) and bool(torch.isfinite(logits).all())
    gw = m2.news_encoder.word_embedding.weight.grad
    touched = set(torch.nonzero(gw.abs().sum(dim=1)).flatten().cpu().tolist())
    words = set()
    for k, mk in (('user_title_text', 'user_title_mask'), ('user_content_text', 'user_content_mask'), ('news_title_text', 'news_title_mask'),
                  ('news_content_text', 'news_content_mask')):
        m = batch[mk].copy()
        m[..., 0] = True                                   # the reference's mask[:, 0] = 1 fix (newsEncoders.py:108-109)
        words |= set(batch[k][m].tolist())
    assert touched <= words and len(touched) >= 0.95 * len(words - {0})
    # embedding-dropout keep rate through the gather kernel
    idx = torch.randint(2, V, (20000,), dtype=torch.int32, device='cuda')
    table = m2.news_encoder.word_embedding.weight.detach()
    kept = ops.embed_gather(table, idx, 0.1, 12345)
    full = ops.embed_gather(table, idx, 0.0, 12345)
    rate = float((kept != 0).float().sum() / (full != 0).float().sum())
    assert abs(rate - 0.9) < 5e-3, rate
    assert torch.allclose(kept[kept != 0], (full / 0.9)[kept != 0], rtol=1e-6)


def test_native_rccl_exchange_single_rank():
    """nnr_dp_* (RCCL called through the C-ABI): a one-rank communicator on this GPU -- all-reduce and broadcast are the
    identity, on the launch stream, and the communicator is created / destroyed cleanly.  (Multi-rank: driver's scaling run.)"""
    from nnr_amd.dp import NativeExchange
    nx = NativeExchange(0, 1)
    x = torch.randn(1 << 20, device='cuda')
    y = x.clone()
    nx.allreduce(y)
    nx.broadcast(y, 0)
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    nx.close()


@pytest.mark.parametrize('binding', ['torch', 'native'])
def test_gradient_exchange_overlaps_the_news_encoder_backward(binding):
    """DDP overlaps its bucketed all-reduce with loss.backward() (trainer.py:297).  Here: the user encoder's gradient bucket goes
    to RCCL when the SUE backward returns -- with the whole news-encoder backward (recurrence + token GEMMs) still ahead -- and
    the rest after the streams are joined.  >= 2 ranks cannot run on this 1-GPU box, so the ORDER is checked on a one-rank RCCL
    communicator (sum over one rank = identity): the early bucket is issued milliseconds before the exchange finishes, and the
    first step's gradients equal the exchange-free trainer's."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['MASTER_PORT'] = str(29533 + (binding == 'native'))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'dp_overlap_check.py'), binding], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    print(out)
    assert out['backward_left_when_early_bucket_went_out_ms'] >= 1.0, out           # the CNE backward of a batch-32 step is several ms
    assert out['max_grad_diff_vs_no_exchange_rel'] <= 1e-5, out                      # (f32 atomics reorder sums between runs)
    assert [b['name'] for b in out['buckets']['buckets']] == ['early (user encoder)', 'table (word embedding)', 'late']
    # the word-embedding table's bucket goes out behind the last embedding-row scatter GEMM, before the tail of the step (the LSTM
    # weight-gradient GEMMs, the stream joins) is done
    assert out['table_bucket_issued_ms'] is not None and out['early_bucket_issued_ms'] < out['table_bucket_issued_ms'] < out['exchange_finished_ms'], out
    assert out['step_left_when_table_bucket_went_out_ms'] >= 0.05, out


def test_bench_launcher_runs_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` as the driver types it (launcher -> torch.distributed.run -> two ranks), in the shared-GPU gloo test mode."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(NNR_DP_BACKEND='gloo', NNR_SHARE_GPU='1', NNR_LSTM_PAIR='0')      # (two processes on one GPU: one-CU recurrence, see tests/test_hip_dp_gpu.py)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no_cpu_baseline', '--batch_size', '8'],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    # headline = the reference's batch semantics (--batch_size is the GLOBAL batch, trainer.py:218); weak scaling is the secondary object
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong' and out['config']['global_batch'] == 8 and out['config']['per_gpu_batch'] == 4 and out['value'] > 0
    assert 'bs=8' in out['metric'] and out['weak_scaling']['per_gpu_batch'] == 8 and out['weak_scaling']['global_batch'] == 16
    assert out['config']['recurrence_exchange_timeouts'] == 0 and 'device-resident corpus' in out['config']['batches']
    assert out['config']['gradient_exchange']['binding'].startswith('torch.distributed')         # gloo test mode: never the RCCL binding
