"""Train mode with dropout ON, HIP path vs the CPU oracle, at sizes the oracle finishes in seconds (the headline-size versions
are in tests/test_hip_headline_gpu.py).  The keep-mask of every dropout site of the call comes from the HIP generator and is
injected into the oracle (tests/hip_masks.py); the oracle's own dropout sites are pinned to the reference by
tests/test_oracle_golden.py::test_oracle_dropout_on_matches_reference.  Covered here: the ragged / empty / maximum-length edge
batch, both call forms of CNE (candidate + history call as one packed token stream, and the plugin API's two separate calls,
each with its own seed), --gcn_layer_norm (dropout fused into the LayerNorm kernel), --no_gcn_residual, gcn_layer_num 1 (no GCN
dropout at all: layers.py:299-300), CNN+ATT at the reference's MIND-small rate 0.25 and MHSA+MHSA (0.2 + the p = 0.5 site).
Reference sites: newsEncoders.py:53,117-118,163,165,193,196; userEncoders.py:80,91,171; layers.py:319-322."""
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


def _compare_dropout_on(model, ref, batch, union=None, tol=1e-4):
    import hip_masks
    from nnr_amd.model import negative_log_softmax
    from oracle import nnr_oracle as O
    for p in model.parameters():
        p.grad = None
    dev = to_torch(batch, 'cuda')
    rates = hip_masks.inject(model, ref, dict(zip(BATCH_FIELDS, dev)), union=union)
    logits = model(*dev)
    loss = negative_log_softmax(logits)
    loss.backward()
    torch.cuda.synchronize()
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
    return rates


def _cne_cfg(**kw):
    kw.setdefault('dropout_rate', 0.2)
    return make_config(['--news_encoder=CNE', '--user_encoder=SUE'], corpus_sizes=dict(vocabulary_size=800), tie_order='stable', **kw)


def test_cne_sue_dropout_on_edge_batch():
    from test_hip_edge_gpu import _edge_batch
    cfg = _cne_cfg(batch_size=4)
    model, ref = _models(cfg)
    rates = _compare_dropout_on(model, ref, _edge_batch(cfg))
    assert abs(rates['sue/affine'] - 0.8) < 0.01 and abs(rates['gcn/0'] - 0.9) < 0.01


def test_cne_sue_dropout_on_two_steps_use_fresh_masks():
    """The per-call seed advances: a second step on the same batch draws different masks, and still matches the oracle."""
    cfg = _cne_cfg(batch_size=4)
    model, ref = _models(cfg, seed=1)
    batch = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size, news_pool=500, seed=3)).batch(4, np.random.default_rng(1))
    _compare_dropout_on(model, ref, batch)
    first = {k: v.clone() for k, v in ref.user_encoder.forced_keep.items()}
    _compare_dropout_on(model, ref, batch)
    assert not torch.equal(first['affine'], ref.user_encoder.forced_keep['affine'])


def test_cne_plugin_api_two_separate_calls_dropout_on(monkeypatch):
    """NNR_CNE_UNION=0 / the plugin API: candidate call and history call are separate encoder calls with separate seeds."""
    from nnr_amd import news_encoders as NE
    monkeypatch.setattr(NE, '_CNE_UNION', False)
    cfg = _cne_cfg(batch_size=3)
    model, ref = _models(cfg, seed=2)
    batch = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size, news_pool=500, seed=4)).batch(3, np.random.default_rng(2))
    _compare_dropout_on(model, ref, batch, union=False)


@pytest.mark.parametrize('flags', [dict(gcn_layer_norm=True), dict(no_gcn_residual=True), dict(gcn_layer_num=1), dict(gcn_layer_num=2, dropout_rate=0.5)])
def test_sue_gcn_variants_dropout_on(flags):
    cfg = _cne_cfg(batch_size=3, **flags)
    model, ref = _models(cfg, seed=3)
    batch = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size, news_pool=500, seed=5)).batch(3, np.random.default_rng(3))
    _compare_dropout_on(model, ref, batch)


@pytest.mark.parametrize('ne,ue,ds', [('CNN', 'ATT', 'small'), ('MHSA', 'MHSA', '200k'), ('MHSA', 'ATT', 'large'), ('CNN', 'MHSA', '200k')])
def test_dense_encoders_dropout_on(ne, ue, ds):
    """CNN+ATT at MIND-small's 0.25 (BASELINE.json configs[0], config.py:84-86), MHSA+MHSA at 0.2 + the p = 0.5 site."""
    cfg = make_config(['--news_encoder=' + ne, '--user_encoder=' + ue, '--dataset=' + ds], corpus_sizes=dict(vocabulary_size=800))
    assert cfg.dropout_rate == {'small': 0.25, '200k': 0.2, 'large': 0.1}[ds]
    model, ref = _models(cfg, seed=4)
    batch = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size, news_pool=500, seed=6)).batch(16 if ne == 'CNN' else 6, np.random.default_rng(4))
    _compare_dropout_on(model, ref, batch)


def test_site_masks_equal_the_kernels_own_masks():
    """tests/hip_masks.py derives each site's mask from nnr_dropout (flat element index).  Here the kernels that APPLY the masks
    are run on all-ones inputs and must produce exactly those masks: embedding gather (row * E + column over packed rows), small
    embedding rows, SUE's proxy rows, the GCN aggregate epilogue and the GEMM epilogue (row * N + column)."""
    import hip_masks
    from nnr_amd import ops
    dev = 'cuda'
    p, seed = 0.2, 12345
    ones = lambda *s: torch.ones(*s, device=dev)
    rows, E = 777, 300
    got = ops.embed_gather(ones(1, E), torch.zeros(rows, dtype=torch.int32, device=dev), p, seed)
    assert torch.equal(got > 0, hip_masks.flat_keep(rows * E, p, seed).view(rows, E))
    assert torch.allclose(got[got > 0], torch.tensor(1.25, device=dev))
    n, dim = 333, 50
    out = torch.zeros(n, 64, device=dev)
    ops.small_embed_fwd(ones(1, dim), torch.zeros(n, dtype=torch.int32, device=dev), out, 64, p, seed + 3)
    assert torch.equal(out[:, :dim] > 0, hip_masks.flat_keep(n * dim, p, seed + 3).view(n, dim))
    B, Hn, Kc, D = 5, 50, 18, 900
    x0 = torch.empty(B, Hn + Kc, D, device=dev)
    ops.sue_x0_fwd(torch.zeros(B, Hn, D, device=dev), ones(Kc, D), x0, B, Hn, Kc, D, p, seed + 1)
    assert torch.equal(x0[:, Hn:] > 0, hip_masks.flat_keep(B * Kc * D, p, seed + 1).view(B, Kc, D))
    G = Hn + Kc
    graph = torch.zeros(B, G, G, device=dev)
    r, y = torch.empty(B, G, D, device=dev), torch.empty(B, G, D, device=dev)
    ops.gcn_aggregate_fwd(graph, torch.zeros(B, G, D, device=dev), ones(D), None, r, y, B, G, D, True, p / 2, seed + 10)
    assert torch.equal(y > 0, hip_masks.flat_keep(B * G * D, p / 2, seed + 10).view(B, G, D))
    M = B * 5 * (Kc + 1)
    f2 = torch.empty(M, D, device=dev)
    rc = torch.empty(M, D, device=dev)
    ops.gemm(torch.zeros(M, D, device=dev), torch.zeros(D, D, device=dev), f2, M=M, N=D, K=D, lda=D, ldb=D, ldc=D, bias=ones(D), act=ops.ACT_RELU,
             aux_out=rc, ldaux=D, resid=None, drop=(3, p, seed + 2, D))
    assert torch.equal(f2 > 0, hip_masks.flat_keep(M * D, p, seed + 2).view(M, D))
