"""User encoders of the hot path (reference: userEncoders.py) on the HIP kernels.

SUE (userEncoders.py:42-98):
  X0 = [history news reps ; dropout_(proxy nodes)]                                   (sue_x0 kernel)
  L x { X <- dropout(relu(A (X W^T) + b) + X) }   -- the dense GEMM first, then the per-user 68x68 aggregate as a batched
                                                     GEMM whose epilogue applies bias / relu / residual / dropout
  gfeat = (GCN(X0) + X0)[:, :50]
  intra-cluster attention: torch_scatter's scatter_softmax / scatter_sum -> sue_intra kernel (segmented LDS reduction;
  the key projection is evaluated once per history item, not on the N-times expanded tensor -- identical math)
  F <- dropout(relu(F W_c^T + b_c) + F)  (GEMM epilogue) ; inter-cluster ScaledDotProduct attention in GEMV form + pool.
Backward is hand-written against the same kernels; parameter gradients accumulate into `param.grad`."""
import math

import os

import torch
import torch.nn as nn

from . import ops
from .layers import Attention, ScaledDotProduct_CandidateAttention, MultiHeadAttention, GCN, grad_of
from .news_encoders import NewsEncoder


class UserEncoder(nn.Module):
    """userEncoders.py:12-39: holds a reference to the SAME news-encoder instance (:16)."""

    def __init__(self, news_encoder: NewsEncoder, config):
        super().__init__()
        self.news_embedding_dim = news_encoder.news_embedding_dim
        self.news_encoder = news_encoder
        self.auxiliary_loss = None
        self._seed_base = int(getattr(config, 'seed', 0)) * 104723 + 29
        self._calls = 0

    def _next_seed(self):
        self._calls += 1
        return (self._seed_base + 15485863 * self._calls) & 0x7FFFFFFF

    def forward(self, user_title_text, user_title_mask, user_title_entity, user_content_text, user_content_mask, user_content_entity,
                user_category, user_subCategory, user_history_mask, user_history_graph, user_history_category_mask,
                user_history_category_indices, user_embedding, candidate_news_representation):
        raise Exception('Function forward must be implemented at sub-class')


class _SUEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hist, cand, mod, graph, cmask, cidx):
        out, saved = sue_forward(mod, hist.contiguous(), cand.contiguous(), graph, cmask, cidx)
        ctx.mod, ctx.saved = mod, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        dhist, dcand = sue_backward(ctx.mod, ctx.saved, dout.contiguous())
        ctx.saved = None
        return dhist, dcand, None, None, None, None


_SUE_JOIN = os.environ.get('NNR_SUE_JOIN', '0') == '1'      # 0 (default) = this encoder's weight-gradient GEMMs (leaf stream) are joined at the end of the step; 1 = the main
                                                             # stream waits for them when this encoder's backward returns (what a data-parallel early bucket needs: forced there).
                                                             # Round 4, same box: batch 8 3.20-3.21 vs 3.33-3.40 ms (the main stream sat idle for ~200 us behind four 65 us GEMMs),
                                                             # batch 16 4.35-4.40 vs 4.42-4.44, batch 64 10.43-10.49 vs 10.47-10.55 (an earlier A/B at batch 64 alone read neutral)
_GCN_FUSED = os.environ.get('NNR_GCN_FUSED', '1') != '0'      # A/B: dedicated per-user aggregate kernel vs the batched tile GEMM
_SUE_SIDE = os.environ.get('NNR_SUE_SIDE', '1') != '0'        # round 4: the candidate-side projections (inputs: the candidates only) and the candidate gradient's
                                                              # accumulations (read by nobody before the news encoder's backward) leave SUE's dependent chain for a side stream
_SIDE = {}
ops.STREAM_CACHES.append(_SIDE)


def _sue_side(dev):
    if not _SUE_SIDE or ops.ONE_STREAM[0]:
        return None
    key = (dev.type, dev.index)
    if key not in _SIDE:
        _SIDE[key] = ops.new_stream(dev, critical=True)
    return _SIDE[key]


def gcn_forward(gcn, x0, graph, seed0, training):
    """GCN.forward (layers.py:318-323) over GCNLayer.forward (:285-292):  X <- dropout(relu(LN?((A X) W^T + b)) + X).  The dense
    product runs first (X W^T on all B*G rows), then the per-user G x G aggregate as a batched GEMM whose epilogue applies bias /
    relu / residual / dropout -- or, with --gcn_layer_norm, only the bias, followed by the fused LayerNorm kernel."""
    B, G, D = x0.shape
    f32 = dict(device=x0.device, dtype=torch.float32)
    Lg = gcn.num_layers
    xs, rs, lns = [x0], [], []
    x = x0
    for l, layer in enumerate(gcn.gcn_layers):
        y = torch.empty((B, G, D), **f32)
        r = torch.empty((B, G, D), **f32)
        pl = (gcn.dropout_rate if training else 0.0) if l + 1 < Lg else 0.0
        z = ops.linear_fwd(x.view(B * G, D), layer.W.weight)                       # X W^T  (bias goes after the aggregate)
        if getattr(layer, 'layer_norm', False):
            u = torch.empty((B, G, D), **f32)
            ops.gemm(graph, z, u, M=G, N=D, K=G, lda=G, ldb=D, ldc=D, trans_b=True, bias=layer.W.bias, batch=B, strideA=G * G, strideB=G * D,
                     strideC=G * D, tile=2)
            xhat = torch.empty((B, G, D), **f32)
            rstd = torch.empty(B * G, **f32)
            ln = layer.layer_normalization
            ops.layernorm_fwd(u, ln.weight, ln.bias, ln.eps, xhat, rstd, r, x if gcn.residual else None, y, pl, seed0 + l)
            lns.append((xhat, rstd))
        elif _GCN_FUSED and G <= 128 and D % 4 == 0:
            ops.gcn_aggregate_fwd(graph, z, layer.W.bias, x if gcn.residual else None, r, y, B, G, D, True, pl, seed0 + l)
            lns.append(None)
        else:
            ops.gemm(graph, z, y, M=G, N=D, K=G, lda=G, ldb=D, ldc=D, trans_b=True, bias=layer.W.bias, act=ops.ACT_RELU, aux_out=r,
                     ldaux=D, resid=x if gcn.residual else None, ldres=D, drop=(3, pl, seed0 + l, D), batch=B, strideA=G * G,
                     strideB=G * D, strideC=G * D, stride_aux=G * D, stride_res=G * D, tile=2)
            lns.append(None)
        xs.append(y)
        rs.append(r)
        x = y
    return x, dict(xs=xs, rs=rs, lns=lns, seed0=seed0, training=training)


def gcn_backward(gcn, gsv, dy, graph, leaf):
    """Gradient of gcn_forward wrt its input; parameter gradients are accumulated (the weight-gradient GEMMs through `leaf`)."""
    B, G, D = dy.shape
    f32 = dict(device=dy.device, dtype=torch.float32)
    Lg = gcn.num_layers
    for l in range(Lg - 1, -1, -1):
        layer = gcn.gcn_layers[l]
        pl = (gcn.dropout_rate if gsv['training'] else 0.0) if l + 1 < Lg else 0.0
        dS = torch.empty((B, G, D), **f32)
        dx = torch.empty((B, G, D), **f32)
        if _GCN_FUSED and G <= 128 and D % 4 == 0 and gsv['lns'][l] is None:
            # mask + ReLU gradient applied while dY is loaded, then dZ_b = A_b^T dS_b, one launch (csrc/gcn.hip)
            dz = torch.empty((B, G, D), **f32)
            ops.gcn_aggregate_bwd(graph, dy, gsv['rs'][l], dS, dx if gcn.residual else None, dz, B, G, D, pl, gsv['seed0'] + l)
            ops.linear_bwd_data(dz.view(B * G, D), layer.W.weight, out=dx.view(B * G, D), accumulate=bool(gcn.residual))
            leaf(lambda dS=dS, dz=dz, l=l, layer=layer: (ops.bias_grad(dS.view(B * G, D), grad_of(layer.W.bias)),
                                                         ops.linear_bwd_weight(dz.view(B * G, D), gsv['xs'][l].view(B * G, D), grad_of(layer.W.weight))), dS, dz)
            dy = dx
            continue
        ops.relu_drop_bwd(dy, gsv['rs'][l], dS, dx, pl, gsv['seed0'] + l)        # dx = masked dy (residual branch)
        if not gcn.residual:
            ops.fill_zero(dx)            # (a C-ABI call, not dx.zero_(): a torch fill is not part of a recorded launch tape -- round-3 advisor)
        if gsv['lns'][l] is not None:                                              # through the LayerNorm: dS := d(A z + b)
            xhat, rstd = gsv['lns'][l]
            ln = layer.layer_normalization
            du = torch.empty((B, G, D), **f32)
            ops.layernorm_bwd(dS, xhat, rstd, ln.weight, du, grad_of(ln.weight), grad_of(ln.bias))
            dS = du
        dz = torch.empty((B, G, D), **f32)                                   # dZ_b = A_b^T dS_b
        ops.gemm(graph, dS, dz, M=G, N=D, K=G, lda=G, ldb=D, ldc=D, trans_a=True, trans_b=True, batch=B, strideA=G * G, strideB=G * D,
                 strideC=G * D, tile=2)
        ops.linear_bwd_data(dz.view(B * G, D), layer.W.weight, out=dx.view(B * G, D), accumulate=True)
        leaf(lambda dS=dS, dz=dz, l=l, layer=layer: (ops.bias_grad(dS.view(B * G, D), grad_of(layer.W.bias)),
                                                     ops.linear_bwd_weight(dz.view(B * G, D), gsv['xs'][l].view(B * G, D), grad_of(layer.W.weight))), dS, dz)
        dy = dx
    return dy


class _GCNFunction(torch.autograd.Function):
    """GCN.forward as a standalone layer (layers.GCN.forward)."""

    @staticmethod
    def forward(ctx, feature, gcn, graph):
        seed = (getattr(gcn, '_calls', 0) * 7919 + 101) & 0x7FFFFFFF
        gcn.__dict__['_calls'] = getattr(gcn, '_calls', 0) + 1
        out, gsv = gcn_forward(gcn, feature.contiguous(), graph.contiguous(), seed, gcn.training)
        ctx.gcn, ctx.gsv, ctx.graph = gcn, gsv, graph.contiguous()
        return out

    @staticmethod
    def backward(ctx, dy):
        with ops.leaf_scope(dy.device, enable=False) as leaf:
            dx = gcn_backward(ctx.gcn, ctx.gsv, dy.contiguous(), ctx.graph, leaf)
        ctx.gsv = None
        return dx, None, None


CAPTURE = [None]      # diagnostics / parity tests: set CAPTURE[0] = [] and every sue_forward appends its saved state (ReLU outputs of the GCN
                      # layers in sv['gcn']['rs'], of the cluster affine in sv['rc']); a replayed step rewrites the SAME buffers


def sue_forward(mod, hist, cand, graph, cmask, cidx):
    B, Hn, D = hist.shape
    N = cand.shape[1]
    Kc = mod.proxy_node_embedding.shape[0]
    G, Cn, A = Hn + Kc, Kc + 1, mod.attention_dim
    dev = hist.device
    f32 = dict(device=dev, dtype=torch.float32)
    p = mod.dropout_rate if mod.training else 0.0
    seed = mod._next_seed()
    sv = dict(B=B, Hn=Hn, D=D, N=N, Kc=Kc, G=G, Cn=Cn, A=A, p=p, seed=seed, hist=hist, cand=cand, cidx=cidx, cmask=cmask,
              graph=graph)
    x0 = torch.empty((B, G, D), **f32)
    # ---- the candidate-side projections need nothing but `cand`: three small launches (q of the intra-cluster attention, q and v of the
    # inter-cluster attention; 15-30 us each, latency-bound) that sat on the dependent chain behind the GCN run beside it instead
    cand2 = cand.view(B * N, D)
    ia = mod.interClusterAttention
    qc = torch.empty((B * N, A), **f32)
    qv = torch.empty((B * N, A), **f32)
    v = torch.empty((B * N, D), **f32)

    def cand_side():
        ops.linear_fwd(cand2, mod.intraCluster_Q.weight, mod.intraCluster_Q.bias, out=qc)                # [B*N, A]
        ops.linear_fwd(cand2, ia.Q.weight, ia.Q.bias, out=qv)                                            # [B*N, A]
        ops.gemm(qv, ia.K.weight, v, M=B * N, N=D, K=A, lda=A, ldb=D, ldc=D, trans_b=True)
    side = _sue_side(dev)
    side_ev = None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            cand_side()
            side_ev = torch.cuda.Event()
            side_ev.record()
    # (+ the in-place `user_history_category_mask[:, -1] = 1` of userEncoders.py:73 in the same launch)
    fix = cmask if (cmask.is_cuda and cmask.dim() == 2 and cmask.is_contiguous() and tuple(cmask.shape) == (B, Kc + 1)) else None
    if fix is None:
        if cmask.is_cuda and cmask.dim() == 2 and cmask.is_contiguous() and cmask.element_size() == 1:
            ops.fill_column_u8(cmask, -1, 1)            # (a C-ABI call: recordable)
        else:
            cmask[:, -1] = 1
    ops.sue_x0_fwd(hist, mod.proxy_node_embedding, x0, B, Hn, Kc, D, p, seed + 1, cmask_fix=fix)
    # ---- GCN
    x, gsv = gcn_forward(mod.gcn, x0, graph, seed + 10, mod.training)
    sv['gcn'] = gsv
    gfeat = torch.empty((B, Hn, D), **f32)
    ops.sue_slice_fwd(x, x0, gfeat, B, Hn, G, D)
    # ---- intra-cluster attention
    kf = ops.linear_fwd(gfeat.view(B * Hn, D), mod.intraCluster_K.weight)                                # [B*Hn, A]
    if side is None:
        cand_side()
    else:
        torch.cuda.current_stream(dev).wait_event(side_ev)
    alpha_i = torch.empty((B, N, Hn), **f32)
    feat = torch.empty((B * N * Cn, D), **f32)
    ops.sue_intra_fwd(kf, qc, gfeat, cidx, B, N, Hn, Cn, A, D, alpha_i, feat)
    # ---- cluster feature affine: dropout(relu(W F + b) + F)
    rc = torch.empty((B * N * Cn, D), **f32)
    f2 = torch.empty((B * N * Cn, D), **f32)
    ops.gemm(feat, mod.clusterFeatureAffine.weight, f2, M=B * N * Cn, N=D, K=D, lda=D, ldb=D, ldc=D, bias=mod.clusterFeatureAffine.bias,
             act=ops.ACT_RELU, aux_out=rc, ldaux=D, resid=feat, ldres=D, drop=(3, p, seed + 2, D))
    # ---- inter-cluster attention (layers.py:196-203) in GEMV form
    alpha_o = torch.empty(B * N * Cn, **f32)
    out = torch.empty((B * N, D), **f32)
    ops.pool_fwd(x=f2, ldx=D, D=D, n=B * N, Lx=Cn, mask=cmask, mask_div=N, v=v, ldv=D, scale=1.0 / math.sqrt(A), alpha=alpha_o, out=out,
                 ldo=D)
    sv.update(gfeat=gfeat, kf=kf, qc=qc, alpha_i=alpha_i, feat=feat, rc=rc, f2=f2, qv=qv, v=v, alpha_o=alpha_o)
    if CAPTURE[0] is not None:
        CAPTURE[0].append(sv)
    return out.view(B, N, D), sv


def sue_backward(mod, sv, dout, dhist_out=None, dcand_accum=None):
    """dhist_out [B, Hn, D] (optional): where the history gradient is written; dcand_accum [B*N, D] (optional): the candidate
    gradient is ADDED into it (the step without autograd sums the click predictor's and this encoder's share there)."""
    B, Hn, D, N, Kc, G, Cn, A, p, seed = (sv[k] for k in ('B', 'Hn', 'D', 'N', 'Kc', 'G', 'Cn', 'A', 'p', 'seed'))
    dev = dout.device
    f32 = dict(device=dev, dtype=torch.float32)
    cand2 = sv['cand'].view(B * N, D)
    dout = dout.view(B * N, D)
    ia = mod.interClusterAttention
    hook = mod.__dict__.get('_grads_ready_hook')
    # a data-parallel trainer starts reducing this encoder's gradients right after this function (early bucket): then they must be
    # ordered on the current stream when it returns.  Otherwise nobody needs them before the optimizer: the leaf stream is joined at
    # the end of the backward pass, and the news encoder's backward starts without waiting for the last weight-gradient GEMMs
    exchange = getattr(hook, '__self__', None)
    need_now = _SUE_JOIN or (exchange is not None and exchange.active()) or not ops._DEFER.get('step_joins')      # (only the native step ends with its own join)
    with ops.leaf_scope(dev, defer_join=not need_now) as leaf:
        res = _sue_backward_body(mod, sv, dout, leaf, B, Hn, D, N, Kc, G, Cn, A, p, seed, dev, f32, cand2, ia, dhist_out, dcand_accum)
    # (joined form) every parameter gradient of this encoder is now ordered on the current stream:
    # a data-parallel trainer starts reducing them while the news encoder's backward is still to come (dp.GradientExchange).
    # (Measured and rejected: issuing this encoder's weight-gradient GEMMs only after its data-gradient chain, so that they
    # overlap the news encoder's backward prologue instead of slowing the 4 352-row chain: 13.05 vs 12.84 ms/step.)
    if hook is not None:
        hook()
    return res


def _sue_backward_body(mod, sv, dout, leaf, B, Hn, D, N, Kc, G, Cn, A, p, seed, dev, f32, cand2, ia, dhist_out=None, dcand_accum=None):
    # ---- inter-cluster pool
    df2 = torch.empty((B * N * Cn, D), **f32)
    dv = torch.empty((B * N, D), **f32)
    ops.pool_bwd(x=sv['f2'], ldx=D, D=D, n=B * N, Lx=Cn, mask=sv['cmask'], mask_div=N, v=sv['v'], ldv=D, scale=1.0 / math.sqrt(A),
                 alpha=sv['alpha_o'], dout=dout, lddo=D, dx=df2, lddx=D, dv=dv, lddv=D)
    dqv = torch.empty((B * N, A), **f32)
    # the step without autograd: the candidate gradient goes straight into the union gradient buffer's candidate rows
    dcand = dcand_accum if dcand_accum is not None else torch.empty((B * N, D), **f32)
    main = torch.cuda.current_stream(dev)
    side = _sue_side(dev)

    def on_side(fn):
        # nobody reads the candidate gradient (or d q_v) before the news encoder's backward: off the chain, joined when this function returns
        if side is None:
            return fn()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            return fn()

    def inter_q():
        ops.gemm(dv, ia.K.weight, dqv, M=B * N, N=A, K=D, lda=D, ldb=D, ldc=A)
        leaf(lambda: (ops.linear_bwd_weight(sv['qv'], dv, grad_of(ia.K.weight)),
                      ops.linear_bwd_weight(dqv, cand2, grad_of(ia.Q.weight), db=grad_of(ia.Q.bias))), dv, dqv)
        ops.linear_bwd_data(dqv, ia.Q.weight, out=dcand, accumulate=dcand_accum is not None)             # [B*N, D]
    on_side(inter_q)
    # ---- cluster affine
    dS = torch.empty((B * N * Cn, D), **f32)
    dfeat = torch.empty((B * N * Cn, D), **f32)
    ops.relu_drop_bwd(df2, sv['rc'], dS, dfeat, p, seed + 2)
    ops.linear_bwd_data(dS, mod.clusterFeatureAffine.weight, out=dfeat, accumulate=True)
    dS_aff = dS
    # (Measured and rejected in round 3: handing the five 900 x 900 weight-gradient GEMMs of this encoder to the news encoder's backward,
    # to run beside the backward recurrence on the 19 KB tile instead of beside this chain's data-gradient GEMMs: 11.41-11.43 vs
    # 11.19-11.25 ms/step.)
    leaf(lambda: ops.linear_bwd_weight(dS_aff, sv['feat'], grad_of(mod.clusterFeatureAffine.weight), db=grad_of(mod.clusterFeatureAffine.bias)), dS_aff)
    # ---- intra-cluster attention
    dg = torch.empty((B, Hn, D), **f32)
    dkf = torch.empty((B * Hn, A), **f32)
    dqc = torch.empty((B * N, A), **f32)
    ops.sue_intra_bwd(sv['kf'], sv['qc'], sv['gfeat'], sv['cidx'], sv['alpha_i'], dfeat, B, N, Hn, Cn, A, D, dg, dkf, dqc)
    ops.linear_bwd_data(dkf, mod.intraCluster_K.weight, out=dg.view(B * Hn, D), accumulate=True)
    on_side(lambda: ops.linear_bwd_data(dqc, mod.intraCluster_Q.weight, out=dcand, accumulate=True))
    leaf(lambda: (ops.linear_bwd_weight(dkf, sv['gfeat'].view(B * Hn, D), grad_of(mod.intraCluster_K.weight)),
                  ops.linear_bwd_weight(dqc, cand2, grad_of(mod.intraCluster_Q.weight), db=grad_of(mod.intraCluster_Q.bias))), dkf, dqc)
    # ---- GCN (+ outer residual)
    dpad = torch.empty((B, G, D), **f32)
    ops.sue_slice_bwd(dg, dpad, B, Hn, G, D)
    dy = gcn_backward(mod.gcn, sv['gcn'], dpad, sv['graph'], leaf)
    dhist = torch.empty((B, Hn, D), **f32) if dhist_out is None else dhist_out
    ops.sue_x0_bwd(dy, dhist, grad_of(mod.proxy_node_embedding), B, Hn, Kc, D, p, seed + 1, dx0_add=dpad)      # d(gcn(X0) + X0)
    if side is not None:
        main.wait_stream(side)                        # the candidate gradient is complete
    return dhist, dcand.view(B, N, D)


class SUE(UserEncoder):
    """userEncoders.py:42-98."""

    def __init__(self, news_encoder: NewsEncoder, config):
        super().__init__(news_encoder, config)
        self.attention_dim = max(config.attention_dim, self.news_embedding_dim // 4)
        self.proxy_node_embedding = nn.Parameter(torch.zeros([config.category_num, self.news_embedding_dim]))
        self.gcn = GCN(in_dim=self.news_embedding_dim, out_dim=self.news_embedding_dim, hidden_dim=self.news_embedding_dim,
                       num_layers=config.gcn_layer_num, dropout=config.dropout_rate / 2, residual=not config.no_gcn_residual,
                       layer_norm=config.gcn_layer_norm)
        self.intraCluster_K = nn.Linear(self.news_embedding_dim, self.attention_dim, bias=False)
        self.intraCluster_Q = nn.Linear(self.news_embedding_dim, self.attention_dim, bias=True)
        self.clusterFeatureAffine = nn.Linear(self.news_embedding_dim, self.news_embedding_dim, bias=True)
        self.interClusterAttention = ScaledDotProduct_CandidateAttention(self.news_embedding_dim, self.news_embedding_dim, self.attention_dim)
        self.dropout_rate = float(config.dropout_rate)
        self.category_num = config.category_num + 1     # extra one category index for padding news
        self.max_history_num = config.max_history_num
        self.attention_scalar = math.sqrt(float(self.attention_dim))

    def initialize(self):
        self.gcn.initialize()
        nn.init.zeros_(self.proxy_node_embedding)
        nn.init.xavier_uniform_(self.intraCluster_K.weight)
        nn.init.xavier_uniform_(self.intraCluster_Q.weight)
        nn.init.zeros_(self.intraCluster_Q.bias)
        nn.init.xavier_uniform_(self.clusterFeatureAffine.weight, gain=nn.init.calculate_gain('relu'))
        nn.init.zeros_(self.clusterFeatureAffine.bias)
        self.interClusterAttention.initialize()

    def forward(self, user_title_text, user_title_mask, user_title_entity, user_content_text, user_content_mask, user_content_entity,
                user_category, user_subCategory, user_history_mask, user_history_graph, user_history_category_mask,
                user_history_category_indices, user_embedding, candidate_news_representation):
        history_embedding = self.news_encoder(user_title_text, user_title_mask, user_title_entity, user_content_text, user_content_mask,
                                              user_content_entity, user_category, user_subCategory, user_embedding)
        return self.encode_user(history_embedding, user_history_mask, user_history_graph, user_history_category_mask,
                                user_history_category_indices, candidate_news_representation)

    def encode_user(self, history_embedding, user_history_mask, user_history_graph, user_history_category_mask,
                    user_history_category_indices, candidate_news_representation):
        """Everything of forward() after the history news have been encoded (userEncoders.py:73-75, 79-98)."""
        # (user_history_category_mask[:, -1] = 1, in place on the caller's tensor, userEncoders.py:73: done by sue_forward's first launch)
        graph = user_history_graph.contiguous()
        cidx = user_history_category_indices.contiguous()
        assert cidx.dtype == torch.int64 and graph.dtype == torch.float32
        return _SUEFunction.apply(history_embedding, candidate_news_representation, self, graph, user_history_category_mask, cidx)

class MHSA(UserEncoder):
    """userEncoders.py:151-173.  Note F.dropout's default p = 0.5 (not dropout_rate) at :171 and the UNMASKED pool at :172."""

    def __init__(self, news_encoder: NewsEncoder, config):
        super().__init__(news_encoder, config)
        self.head_num, self.head_dim, self.max_history_num = config.head_num, config.head_dim, config.max_history_num
        self.multiheadAttention = MultiHeadAttention(config.head_num, self.news_embedding_dim, config.max_history_num,
                                                     config.max_history_num, config.head_dim, config.head_dim)
        self.affine = nn.Linear(config.head_num * config.head_dim, self.news_embedding_dim, bias=True)
        self.attention = Attention(self.news_embedding_dim, config.attention_dim)

    def initialize(self):
        self.multiheadAttention.initialize()
        nn.init.xavier_uniform_(self.affine.weight, gain=nn.init.calculate_gain('relu'))
        nn.init.zeros_(self.affine.bias)
        self.attention.initialize()

    def forward(self, user_title_text, user_title_mask, user_title_entity, user_content_text, user_content_mask, user_content_entity,
                user_category, user_subCategory, user_history_mask, user_history_graph, user_history_category_mask,
                user_history_category_indices, user_embedding, candidate_news_representation):
        history_embedding = self.news_encoder(user_title_text, user_title_mask, user_title_entity, user_content_text, user_content_mask,
                                              user_content_entity, user_category, user_subCategory, user_embedding)
        return self.encode_user(history_embedding, user_history_mask, user_history_graph, user_history_category_mask,
                                user_history_category_indices, candidate_news_representation)

    def encode_user(self, history_embedding, user_history_mask, user_history_graph, user_history_category_mask,
                    user_history_category_indices, candidate_news_representation):
        from . import functional as Fn
        news_num = candidate_news_representation.size(1)
        B, Hn, D = history_embedding.shape
        qkv = Fn.QKVFn.apply(history_embedding.reshape(B * Hn, D), self.multiheadAttention, None)
        h = Fn.MhsaCoreFn.apply(qkv, user_history_mask.contiguous(), B, Hn, self.head_num, self.head_dim)
        # relu(dropout(affine(h))) == dropout(relu(affine(h))): the mask scales by a non-negative factor
        h = Fn.LinearFn.apply(h, self.affine.weight, self.affine.bias, ops.ACT_RELU, 0.5 if self.training else 0.0, self._next_seed())
        user = self.attention(h.view(B, Hn, D))                                                            # unmasked
        return Fn.ExpandFn.apply(user, news_num)

class ATT(UserEncoder):
    """userEncoders.py:176-191: unmasked additive attention over the history slots (padded slots participate)."""

    def __init__(self, news_encoder: NewsEncoder, config):
        super().__init__(news_encoder, config)
        self.attention = Attention(self.news_embedding_dim, config.attention_dim)

    def initialize(self):
        self.attention.initialize()

    def forward(self, user_title_text, user_title_mask, user_title_entity, user_content_text, user_content_mask, user_content_entity,
                user_category, user_subCategory, user_history_mask, user_history_graph, user_history_category_mask,
                user_history_category_indices, user_embedding, candidate_news_representation):
        history_embedding = self.news_encoder(user_title_text, user_title_mask, user_title_entity, user_content_text, user_content_mask,
                                              user_content_entity, user_category, user_subCategory, user_embedding)
        return self.encode_user(history_embedding, user_history_mask, user_history_graph, user_history_category_mask,
                                user_history_category_indices, candidate_news_representation)

    def encode_user(self, history_embedding, user_history_mask, user_history_graph, user_history_category_mask,
                    user_history_category_indices, candidate_news_representation):
        from . import functional as Fn
        return Fn.ExpandFn.apply(self.attention(history_embedding), candidate_news_representation.size(1))
