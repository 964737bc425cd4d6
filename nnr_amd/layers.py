"""Layer classes of the hot path with the reference's names, constructor signatures and parameter names
(layers.py of the reference), so checkpoints / state_dicts are interchangeable.  Inside CNE / SUE the encoders drive
the HIP kernels directly from these parameter holders; the standalone `forward`s (dense [n, L, F] inputs, used by the
MHSA / ATT / CNN encoders) are autograd Functions over the same kernels."""
import math

import torch
import torch.nn as nn

from . import ops


# Bumped by every optimizer step that updates parameters through raw pointers (nnr_amd.trainer): derived weight layouts
# (e.g. the packed LSTM fragments) are cached against it, together with torch's own version counters.
PARAM_EPOCH = [0]


def grad_of(p):
    """`p.grad`, created zero-filled on first use.  The hand-written backward passes ACCUMULATE into it (the news
    encoder runs twice per step), exactly like autograd's AccumulateGrad; the trainer zeroes the flat gradient buffer."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def stacked_view(ts):
    """One [len(ts)*rows, ...] view over tensors that sit back to back in the same storage (see adjacent_parameter_groups),
    or None when they do not (a model that was not re-homed by the trainer: the callers then take the per-tensor path)."""
    t0 = ts[0]
    if not t0.is_contiguous():
        return None
    step = t0.numel()
    for i, t in enumerate(ts):
        if t.shape != t0.shape or not t.is_contiguous() or t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr() \
                or t.storage_offset() != t0.storage_offset() + i * step:
            return None
    shape = (len(ts) * t0.shape[0],) + tuple(t0.shape[1:])
    stride = t0.stride()
    return t0.as_strided(shape, stride)


class LSTMParams(nn.Module):
    """Parameter holder with nn.LSTM(bidirectional=True, num_layers=1) names/shapes/default init (newsEncoders.py:66-67)."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        k = 1.0 / math.sqrt(hidden_dim)
        for sfx in ('', '_reverse'):
            for name, shape in (('weight_ih_l0', (4 * hidden_dim, input_dim)), ('weight_hh_l0', (4 * hidden_dim, hidden_dim)),
                                ('bias_ih_l0', (4 * hidden_dim,)), ('bias_hh_l0', (4 * hidden_dim,))):
                self.register_parameter(name + sfx, nn.Parameter(torch.empty(shape).uniform_(-k, k)))

    def param_list(self):
        return [self.weight_ih_l0, self.weight_hh_l0, self.bias_ih_l0, self.bias_hh_l0,
                self.weight_ih_l0_reverse, self.weight_hh_l0_reverse, self.bias_ih_l0_reverse, self.bias_hh_l0_reverse]


# ------------------------------------------------------------------------------------------------ additive attention
class _AttentionFn(torch.autograd.Function):
    """layers.py:167-175 on a dense [n, L, F] feature: tanh GEMM, w2 row-dot, then the softmax pool."""

    @staticmethod
    def forward(ctx, feature, mod, mask):
        n, Lx, F = feature.shape
        A = mod.affine1.weight.shape[0]
        x = feature.contiguous().view(n * Lx, F)
        f32 = dict(device=x.device, dtype=torch.float32)
        th = torch.empty((n * Lx, A), **f32)
        score = torch.empty(n * Lx, **f32)
        ops.gemm(x, mod.affine1.weight, th, M=n * Lx, N=A, K=F, lda=F, ldb=F, ldc=A, bias=mod.affine1.bias, act=ops.ACT_TANH)
        ops.rowdot(th, mod.affine2.weight, score)
        alpha = torch.empty(n * Lx, **f32)
        out = torch.empty((n, F), **f32)
        ops.pool_fwd(x=x, ldx=F, D=F, n=n, Lx=Lx, mask=mask, score=score, alpha=alpha, out=out, ldo=F)
        ctx.mod, ctx.mask, ctx.saved = mod, mask, (x, th, alpha, n, Lx, F, A)
        return out

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        x, th, alpha, n, Lx, F, A = ctx.saved
        f32 = dict(device=x.device, dtype=torch.float32)
        dx = torch.empty((n * Lx, F), **f32)
        ds = torch.empty(n * Lx, **f32)
        ops.pool_bwd(x=x, ldx=F, D=F, n=n, Lx=Lx, mask=ctx.mask, alpha=alpha, dout=dout.contiguous(), lddo=F, dx=dx, lddx=F, dscore=ds)
        ops.tanh_score_bwd(th, ds, mod.affine2.weight, grad_of(mod.affine2.weight), None, A)
        gw, gb = grad_of(mod.affine1.weight), grad_of(mod.affine1.bias)
        ops.leaf_deferred(x.device, n * Lx, lambda: ops.linear_bwd_weight(th, x, gw, db=gb), th, x)      # leaf: own stream (ops.leaf_deferred)
        if ops.USE_WT and n * Lx >= 1024 and (A & 3) == 0:
            ops.gemm(th, ops.wt(mod.affine1.weight), dx, M=n * Lx, N=F, K=A, lda=A, ldb=A, ldc=F, accumulate=True)      # NT on W1^T
        else:
            ops.gemm(th, mod.affine1.weight, dx, M=n * Lx, N=F, K=A, lda=A, ldb=F, ldc=F, trans_b=True, accumulate=True)
        return dx.view(n, Lx, F), None, None


class Attention(nn.Module):
    """layers.py:151-175."""

    def __init__(self, feature_dim: int, attention_dim: int):
        super().__init__()
        self.affine1 = nn.Linear(feature_dim, attention_dim, bias=True)
        self.affine2 = nn.Linear(attention_dim, 1, bias=False)

    def initialize(self):
        nn.init.xavier_uniform_(self.affine1.weight, gain=nn.init.calculate_gain('tanh'))
        nn.init.zeros_(self.affine1.bias)
        nn.init.xavier_uniform_(self.affine2.weight)

    def forward(self, feature, mask=None):
        return _AttentionFn.apply(feature, self, mask)


class ScaledDotProduct_CandidateAttention(nn.Module):
    """layers.py:178-203 (parameter holder; CNE and SUE evaluate it in its GEMV form, see their pipelines)."""

    def __init__(self, feature_dim: int, query_dim: int, attention_dim: int):
        super().__init__()
        self.K = nn.Linear(feature_dim, attention_dim, bias=False)
        self.Q = nn.Linear(query_dim, attention_dim, bias=True)
        self.attention_scalar = math.sqrt(float(attention_dim))

    def initialize(self):
        nn.init.xavier_uniform_(self.K.weight)
        nn.init.xavier_uniform_(self.Q.weight)
        nn.init.zeros_(self.Q.bias)

    def forward(self, feature, query, mask=None):
        """layers.py:196-203 on its own: feature [n, L, F], query [n, Qd], mask [n, L] -> [n, F].  GEMV form: the score of
        position t is <feature_t, K^T (Q query + b)> / sqrt(A), so K is never applied to the L x F features."""
        return _CandidateAttentionFn.apply(feature, query, self, mask)


class _CandidateAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feature, query, mod, mask):
        n, Lx, F = feature.shape
        A = mod.K.weight.shape[0]
        x = feature.contiguous().view(n * Lx, F)
        q = query.contiguous()
        f32 = dict(device=x.device, dtype=torch.float32)
        qv = ops.linear_fwd(q, mod.Q.weight, mod.Q.bias)                                       # [n, A]
        v = torch.empty((n, F), **f32)
        ops.gemm(qv, mod.K.weight, v, M=n, N=F, K=A, lda=A, ldb=F, ldc=F, trans_b=True)         # K^T (Q q + b)
        alpha = torch.empty(n * Lx, **f32)
        out = torch.empty((n, F), **f32)
        ops.pool_fwd(x=x, ldx=F, D=F, n=n, Lx=Lx, mask=mask, v=v, ldv=F, scale=1.0 / math.sqrt(A), alpha=alpha, out=out, ldo=F)
        ctx.mod, ctx.mask, ctx.saved = mod, mask, (x, q, qv, v, alpha, n, Lx, F, A)
        return out

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        x, q, qv, v, alpha, n, Lx, F, A = ctx.saved
        f32 = dict(device=x.device, dtype=torch.float32)
        dx = torch.empty((n * Lx, F), **f32)
        dv = torch.empty((n, F), **f32)
        ops.pool_bwd(x=x, ldx=F, D=F, n=n, Lx=Lx, mask=ctx.mask, v=v, ldv=F, scale=1.0 / math.sqrt(A), alpha=alpha, dout=dout.contiguous(),
                     lddo=F, dx=dx, lddx=F, dv=dv, lddv=F)
        dqv = torch.empty((n, A), **f32)
        ops.gemm(dv, mod.K.weight, dqv, M=n, N=A, K=F, lda=F, ldb=F, ldc=A)                     # dqv = dv . K^T
        ops.linear_bwd_weight(qv, dv, grad_of(mod.K.weight))                                    # dK[A, F] += qv^T dv
        ops.linear_bwd_weight(dqv, q, grad_of(mod.Q.weight))
        ops.bias_grad(dqv, grad_of(mod.Q.bias))
        dq = ops.linear_bwd_data(dqv, mod.Q.weight)
        return dx.view(n, Lx, F), dq, None, None


class MultiHeadAttention(nn.Module):
    """layers.py:102-148 (parameter holder + forward over the MFMA attention kernel, see news_encoders.MHSA)."""

    def __init__(self, h: int, d_model: int, len_q: int, len_k: int, d_k: int, d_v: int):
        super().__init__()
        self.h, self.d_model, self.len_q, self.len_k, self.d_k, self.d_v = h, d_model, len_q, len_k, d_k, d_v
        self.out_dim = h * d_v
        self.attention_scalar = math.sqrt(float(d_k))
        self.W_Q = nn.Linear(d_model, h * d_k, bias=True)
        self.W_K = nn.Linear(d_model, h * d_k, bias=True)
        self.W_V = nn.Linear(d_model, h * d_v, bias=True)

    def initialize(self):
        nn.init.xavier_uniform_(self.W_Q.weight)
        nn.init.zeros_(self.W_Q.bias)
        nn.init.xavier_uniform_(self.W_K.weight)
        nn.init.zeros_(self.W_K.bias)
        nn.init.xavier_uniform_(self.W_V.weight)
        nn.init.zeros_(self.W_V.bias)

    def adjacent_parameter_groups(self):
        """Parameters the flat buffer (trainer.FlatParams) should lay out back to back, in this order: the three projection
        weights then form ONE [3*h*d, d_model] matrix (and the biases one vector) without any copy, and functional.QKVFn runs
        one GEMM per direction instead of three.  Names, shapes and the state_dict stay the reference's."""
        if self.W_Q.weight.shape != self.W_K.weight.shape or self.W_Q.weight.shape != self.W_V.weight.shape:
            return []
        return [[self.W_Q.weight, self.W_K.weight, self.W_V.weight], [self.W_Q.bias, self.W_K.bias, self.W_V.bias]]


class Conv1D(nn.Module):
    """layers.py:7-44, 'naive' branch (the only one the in-scope CNN encoder uses)."""

    def __init__(self, cnn_method: str, in_channels: int, cnn_kernel_num: int, cnn_window_size: int):
        super().__init__()
        assert cnn_method == 'naive', 'only cnn_method=naive is on the hot path (SURVEY.md section 2, row 5)'
        self.cnn_method = cnn_method
        self.conv = nn.Conv1d(in_channels=in_channels, out_channels=cnn_kernel_num, kernel_size=cnn_window_size,
                              padding=(cnn_window_size - 1) // 2)


class GCNLayer(nn.Module):
    """layers.py:265-292 (parameter holder; SUE's pipeline evaluates relu(A (X W^T) + b) + X)."""

    def __init__(self, in_dim, out_dim, residual=False, layer_norm=False):
        super().__init__()
        if residual and in_dim != out_dim:
            raise Exception('To facilitate residual connection, in_dim must equal to out_dim')
        self.residual = residual
        self.layer_norm = layer_norm
        self.W = nn.Linear(in_dim, out_dim, bias=True)
        if self.layer_norm:
            self.layer_normalization = nn.LayerNorm(normalized_shape=[out_dim])       # layers.py:273-274

    def initialize(self):
        nn.init.xavier_uniform_(self.W.weight, gain=nn.init.calculate_gain('relu'))
        nn.init.zeros_(self.W.bias)


class GCN(nn.Module):
    """layers.py:294-323."""

    def __init__(self, in_dim, out_dim, hidden_dim=0, num_layers=1, dropout=0.1, residual=False, layer_norm=False):
        super().__init__()
        assert in_dim == out_dim and (num_layers == 1 or hidden_dim == in_dim)
        self.num_layers = num_layers
        self.dropout_rate = float(dropout)
        self.residual = residual
        self.gcn_layers = nn.ModuleList([GCNLayer(in_dim, in_dim, residual=residual, layer_norm=layer_norm) for _ in range(num_layers)])

    def initialize(self):
        for gcn_layer in self.gcn_layers:
            gcn_layer.initialize()

    def forward(self, feature, graph):
        """layers.py:318-323 on its own (the SUE pipeline drives the same kernels directly): feature [B, G, D], graph [B, G, G]."""
        from .user_encoders import _GCNFunction
        return _GCNFunction.apply(feature, self, graph)
