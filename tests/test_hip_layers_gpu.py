"""Standalone layer forwards of the drop-in surface -- a caller composing the reference's layers outside CNE / SUE gets the same
kernels: `ScaledDotProduct_CandidateAttention.forward` (layers.py:196-203), `GCN.forward` (layers.py:318-323, with and without
--gcn_layer_norm / residual) -- and the fused LayerNorm kernels, each against the oracle's restatement / torch in fp64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, e, tol=2e-5, what=''):
    a, e = a.detach().double().cpu(), e.detach().double().cpu()
    scale = max(1.0, float(e.abs().max()))
    err = float((a - e).abs().max())
    assert err <= tol * scale, '%s: max err %.3e (scale %.3e)' % (what, err, scale)


@pytest.mark.parametrize('rows,D,resid,p', [(4352, 900, True, 0.0), (37, 900, False, 0.0), (130, 64, True, 0.3), (5, 1024, True, 0.0)])
def test_layernorm_relu_residual_dropout_fwd_bwd(rows, D, resid, p):
    from nnr_amd import ops
    d = torch.device('cuda')
    g = torch.Generator().manual_seed(rows + D)
    u = torch.randn(rows, D, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    res = torch.randn(rows, D, generator=g) if resid else None
    dy = torch.randn(rows, D, generator=g)
    f32 = dict(device=d, dtype=torch.float32)
    xhat, rstd, r, y = torch.empty(rows, D, **f32), torch.empty(rows, **f32), torch.empty(rows, D, **f32), torch.empty(rows, D, **f32)
    seed = 1234
    ops.layernorm_fwd(u.to(d), gamma.to(d), beta.to(d), 1e-5, xhat, rstd, r, res.to(d) if resid else None, y, p, seed)
    keep = (ops.dropout(torch.ones(rows * D, device=d), p, seed) > 0).view(rows, D).cpu() if p > 0 else torch.ones(rows, D, dtype=torch.bool)
    ud = u.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    v = torch.nn.functional.layer_norm(ud, [D], gd, bd, 1e-5)
    rr = torch.relu(v)
    out = (rr + (res.double() if resid else 0.0)) * keep.double() / (1.0 - p)
    _close(r, rr, what='relu(LN)')
    _close(y, out, what='LN output')
    out.backward(dy.double())
    # backward as the GCN pipeline composes it: relu_drop_bwd -> layernorm_bwd
    dS, dx = torch.empty(rows, D, **f32), torch.empty(rows, D, **f32)
    ops.relu_drop_bwd(dy.to(d), r, dS, dx, p, seed)
    du, dgam, dbet = torch.empty(rows, D, **f32), torch.zeros(D, **f32), torch.zeros(D, **f32)
    ops.layernorm_bwd(dS, xhat, rstd, gamma.to(d), du, dgam, dbet)
    _close(du, ud.grad, tol=5e-5, what='d LN input')
    _close(dgam, gd.grad, tol=5e-5, what='d gamma')
    _close(dbet, bd.grad, tol=5e-5, what='d beta')


@pytest.mark.parametrize('masked', [True, False])
def test_candidate_attention_standalone_forward_backward(masked):
    from nnr_amd.layers import ScaledDotProduct_CandidateAttention
    from oracle.nnr_oracle import CandidatePool
    torch.manual_seed(3)
    n, Lx, F, Q, A = 37, 19, 900, 900, 225
    ref = CandidatePool(F, Q, A).double()
    ref.initialize()
    with torch.no_grad():
        ref.Q.bias.normal_(0, 0.1)
    mod = ScaledDotProduct_CandidateAttention(F, Q, A)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.cuda()
    x, q = torch.randn(n, Lx, F), torch.randn(n, Q)
    mask = (torch.rand(n, Lx) > 0.3) if masked else None
    if masked:
        mask[:, 0] = True
    dout = torch.randn(n, F)
    xr, qr = x.double().requires_grad_(True), q.double().requires_grad_(True)
    ref(xr, qr, mask).backward(dout.double())
    xg, qg = x.cuda().requires_grad_(True), q.cuda().requires_grad_(True)
    out = mod(xg, qg, mask.cuda() if masked else None)
    _close(out, ref(xr, qr, mask), what='candidate attention')
    out.backward(dout.cuda())
    _close(xg.grad, xr.grad, tol=5e-5, what='d feature')
    _close(qg.grad, qr.grad, tol=5e-5, what='d query')
    for k, p in mod.named_parameters():
        _close(p.grad, dict(ref.named_parameters())[k].grad, tol=5e-5, what='d ' + k)


@pytest.mark.parametrize('layer_norm,residual', [(False, True), (True, True), (True, False)])
def test_gcn_standalone_forward_backward(layer_norm, residual):
    from nnr_amd.layers import GCN
    from oracle import nnr_oracle as O
    torch.manual_seed(5)
    B, G, D, Lg = 6, 68, 900, 3
    ref = O.GCN(D, Lg, 0.0, residual, layer_norm).double()
    ref.initialize()
    with torch.no_grad():
        for l in ref.gcn_layers:
            l.W.bias.normal_(0, 0.05)
            if layer_norm:
                l.layer_normalization.weight.uniform_(0.5, 1.5)
                l.layer_normalization.bias.normal_(0, 0.1)
    mod = GCN(D, D, hidden_dim=D, num_layers=Lg, dropout=0.0, residual=residual, layer_norm=layer_norm)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.cuda().train()
    x = torch.randn(B, G, D) * 0.5
    graph = torch.rand(B, G, G) * (torch.rand(B, G, G) > 0.7) / 8
    dy = torch.randn(B, G, D)
    xr = x.double().requires_grad_(True)
    ref.train()
    yr = ref(xr, graph.double())
    yr.backward(dy.double())
    xg = x.cuda().requires_grad_(True)
    y = mod(xg, graph.cuda())
    _close(y, yr, tol=5e-5, what='GCN forward')
    y.backward(dy.cuda())
    _close(xg.grad, xr.grad, tol=1e-4, what='d feature')
    rp = dict(ref.named_parameters())
    for k, p in mod.named_parameters():
        _close(p.grad, rp[k].grad, tol=1e-4, what='d ' + k)
