"""News encoders of the hot path -- same class names, constructor signatures, parameter names and forward
signatures as the reference's newsEncoders.py, computed by the HIP kernels of libnnr_hip.so.

CNE (newsEncoders.py:57-141) pipeline per token stream (title L=32 / content L=128), all on packed valid tokens only:
  seq_plan  -> [embedding-row gather + dropout + x.W_ih^T + b]  (one GEMM, gather fused in the A loader)
            -> persistent Bi-LSTM recurrence (both streams, both directions, ONE launch)
            -> cross-selective gate  H * sigmoid(H.W_H^T + M(c_other))   (GEMM epilogue)
            -> additive self attention (GEMM with fused tanh.w2 row-dot) -> wave-softmax pool
            -> cross attention  score = <H~_t, K^T(Q q + b)> / sqrt(A)   (GEMV form) -> pool
            -> feature fusion with category / subCategory rows.
The backward pass is written by hand against the same kernels (no autograd graph inside the encoder); parameter
gradients are accumulated straight into `param.grad`.

Tie order (see oracle/nnr_oracle.py:length_order): the reference sorts each stream by length with an unstable sort and
gates the title at title-rank r with the content memory at content-rank r.  Because this implementation keeps each stream
in its own sorted order end to end, that pairing is reproduced by construction; `tie_order` only selects how equal lengths
are ordered: 'stable' (device-side, no host sync; = torch 1.12.1 CPU behaviour) or 'torch' (ask the installed torch on the
host, exactly as the reference's CPU path would -- costs a device->host sync, used for bit-parity tests)."""
import math
import os
import pickle

import torch
import torch.nn as nn

from . import ops
from .layers import Attention, ScaledDotProduct_CandidateAttention, MultiHeadAttention, Conv1D, LSTMParams, grad_of, PARAM_EPOCH

_SITE = dict(title=1, content=2, cat=3, sub=4)
_TITLE_DX_FIRST = os.environ.get('NNR_TITLE_DX_FIRST', '0') == '1'       # measured: no gain either way (12.39 vs 12.42-12.58 ms/step)
_TITLE_DX_TILE = int(os.environ.get('NNR_TITLE_DX_TILE', '0'))
_DP_TABLE_FIRST = int(os.environ.get('NNR_DP_TABLE_FIRST', '0'))       # 1: always, -1: when world_size > 1, 0 (default): never -- no multi-GPU box to measure it on


def _dp_world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_initialized() else 1


_CN_SPLIT = os.environ.get('NNR_CN_SPLIT', '0') == '1'      # c_n projections of a union call as plain skinny GEMM + row gather: measured no gain (11.21-11.27 vs 11.24-11.26 ms)
_DX_TILE = int(os.environ.get('NNR_DX_TILE', '0'))          # A/B: tile of the content streams' embedding-row gradient GEMM (0 = automatic: 9)


class NewsEncoder(nn.Module):
    """newsEncoders.py:11-54."""

    def __init__(self, config, word_table=None):
        super().__init__()
        self.word_embedding_dim = config.word_embedding_dim
        self.word_embedding = nn.Embedding(num_embeddings=config.vocabulary_size, embedding_dim=self.word_embedding_dim)
        fname = 'word_embedding-' + str(config.word_threshold) + '-' + str(config.word_embedding_dim) + '-' + config.tokenizer + '-' + \
                str(config.max_title_length) + '-' + str(config.max_abstract_length) + '-' + config.dataset + '.pkl'
        if word_table is None and os.path.exists(fname):            # the reference's behaviour (newsEncoders.py:16-17)
            with open(fname, 'rb') as f:
                word_table = pickle.load(f)
        if word_table is not None:
            self.word_embedding.weight.data.copy_(word_table)
        self.category_embedding = nn.Embedding(num_embeddings=config.category_num, embedding_dim=config.category_embedding_dim)
        self.subCategory_embedding = nn.Embedding(num_embeddings=config.subCategory_num, embedding_dim=config.subCategory_embedding_dim)
        self.dropout_rate = float(config.dropout_rate)
        self.auxiliary_loss = None
        self._seed_base = int(getattr(config, 'seed', 0)) * 7919 + 17
        self._calls = 0

    def _next_seed(self):
        self._calls += 1
        return (self._seed_base + 104729 * self._calls) & 0x7FFFFFFF

    def initialize(self):
        nn.init.uniform_(self.category_embedding.weight, -0.1, 0.1)
        nn.init.uniform_(self.subCategory_embedding.weight, -0.1, 0.1)
        with torch.no_grad():
            self.subCategory_embedding.weight[0].zero_()

    def forward(self, title_text, title_mask, title_entity, content_text, content_mask, content_entity, category, subCategory, user_embedding):
        raise Exception('Function forward must be implemented at sub-class')


def _i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


# ================================================================================================== CNE
class _CNEPairFunction(torch.autograd.Function):
    """Candidate call + history call of the SAME encoder in one pass.  Default: both calls planned as ONE packed token stream
    (cne union: every per-token kernel runs once over both calls; the rank pairing of newsEncoders.py:128-129 stays per call
    through nnr_cne_pair_map).  NNR_CNE_UNION=0: two lock-step calls that only share the recurrence launches."""

    @staticmethod
    def forward(ctx, anchor, mod, *tensors):
        ctx.mod = mod
        if _CNE_UNION:
            ((rep_a, rep_b), sv), = cne_forward_many(mod, [tuple(zip(tensors[:6], tensors[6:]))])
            ctx.saved = (sv,)
            return rep_a, rep_b
        (rep_a, sv_a), (rep_b, sv_b) = cne_forward_many(mod, [tensors[:6], tensors[6:]])
        ctx.saved = (sv_a, sv_b)
        return rep_a, rep_b

    @staticmethod
    def backward(ctx, drep_a, drep_b):
        if len(ctx.saved) == 1:
            sv, = ctx.saved
            D = drep_a.shape[-1]
            drep = torch.cat([drep_a.reshape(-1, D), drep_b.reshape(-1, D)])
            cne_backward_many(ctx.mod, [(sv, drep)])
        else:
            sv_a, sv_b = ctx.saved
            cne_backward_many(ctx.mod, [(sv_a, drep_a.contiguous()), (sv_b, drep_b.contiguous())])
        ctx.saved = None
        return (None,) * 14


class _CNEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, mod, title_text, title_mask, content_text, content_mask, category, subCategory):
        rep, saved = cne_forward(mod, title_text, title_mask, content_text, content_mask, category, subCategory)
        ctx.mod, ctx.saved = mod, saved
        return rep

    @staticmethod
    def backward(ctx, drep):
        cne_backward(ctx.mod, ctx.saved, drep.contiguous())
        ctx.saved = None
        return (None,) * 8


def cne_forward(mod, title_text, title_mask, content_text, content_mask, category, subCategory):
    """Single encoder call (plugin API).  Model.forward uses cne_forward_many to share the recurrence launch between the
    candidate call and the history call."""
    (rep, saved), = cne_forward_many(mod, [(title_text, title_mask, content_text, content_mask, category, subCategory)])
    return rep, saved


_SIDE = {}
ops.STREAM_CACHES.append(_SIDE)
_LSTM_FWD_SPLIT = os.environ.get('NNR_LSTM_FWD_SPLIT', '0') == '1'      # A/B: title recurrence launched on the title stream
_PROJ_ORDER = os.environ.get('NNR_PROJ_ORDER', '0') == '1'      # A/B (round 4, with NNR_LSTM_FWD_SPLIT=1): the content projection waits for the title projection, so the
                                                                # title recurrence runs UNDER the content projection instead of inside the shared recurrence launch
_DWHH_FIRST = os.environ.get('NNR_DWHH_FIRST', '0') == '1'      # A/B (round 4): this stream's dW_hh GEMM in front of the embedding-row gradient GEMM + scatter instead of behind them
_LEAF2_ROWS = int(os.environ.get('NNR_LEAF2_ROWS', '65536'))
_POOL_FUSED = os.environ.get('NNR_POOL_FUSED', '1') != '0'      # A/B (round 5): the two pools' token gradient in ONE write of dHt (see _cne_bwd_pre)
_POST_INLINE = os.environ.get('NNR_POST_INLINE', '0') == '1'      # A/B: the content stream's tail GEMMs on one HIP stream
_DX_SPLIT = os.environ.get('NNR_DX_SPLIT', '1') == '1'      # embedding-row gradient as plain GEMM + scatter kernel (11.46 vs 11.51 ms fused)
_BWD_SPLIT = os.environ.get('NNR_LSTM_BWD_SPLIT', '1') == '1'      # batch 8: 4.68 (split) vs 4.93 ms; batch 64: no difference
_PROJ_TILE = int(os.environ.get('NNR_PROJ_TILE', '0'))       # A/B (round 4): tile of the LSTM input projection (N = 1664 = 8 x 208), 0 = automatic (15)
_DCN_TILE = int(os.environ.get('NNR_DCN_TILE', '0'))         # round 6: tile of the d c_n product in front of the backward recurrence (0 = automatic: 6)
_GATE_TILE = int(os.environ.get('NNR_GATE_TILE', '0'))       # ... of the gate / attention / their data-gradient GEMMs over the token rows (N = 400 / 200)
_GATE_FUSED = os.environ.get('NNR_GATE_FUSED', '1') != '0'      # A/B: the gate's backward inside the epilogue of the GEMM that completes dHt
_CNE_UNION = os.environ.get('NNR_CNE_UNION', '1') != '0'      # A/B switch: candidate + history call as one packed token stream


def _side_stream(dev):
    """One extra HIP stream per device for the small (candidate) encoder call."""
    key = (dev.type, dev.index)
    if key not in _SIDE:
        _SIDE[key] = ops.new_stream(dev, critical=True)
    return _SIDE[key]


def _title_stream(dev):
    key = (dev.type, dev.index, 2)
    if key not in _SIDE:
        _SIDE[key] = ops.new_stream(dev, critical=True)
    return _SIDE[key]


def _two_chains(dev, enable, title_fn, content_fn):
    """The title and the content chain of ONE encoder call are independent between their exchange points (c_n before the
    gate, the self-attention vectors before the cross attention and their gradients on the way back): on the big call the
    title chain runs on a third HIP stream next to the content chain.  Half of the step at batch 64 (7 of 16 ms) is the
    latency of dependent launches, not throughput; the title kernels are a quarter of the content kernels' size."""
    if not enable:
        title_fn()
        content_fn()
        return
    main = torch.cuda.current_stream(dev)
    s2 = _title_stream(dev)
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        title_fn()
    content_fn()
    main.wait_stream(s2)


class _on:
    """Run a phase of call #0 on the side stream when there are several calls; main-stream phases pass through."""

    def __init__(self, side):
        self.side = side

    def __enter__(self):
        if self.side is not None:
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.side is not None:
            self.ctx.__exit__(*a)


def _fork_join(n_calls, dev, phase):
    """phase(i) for every call: call 0 (the candidates: ~10 % of the rows, launch-latency-bound kernels) on the side stream,
    concurrently with the other calls on the current stream; returns after both streams are joined."""
    if n_calls == 1:
        return [phase(0, True)]
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        first = phase(0, False)
    rest = [phase(i, True) for i in range(1, n_calls)]
    main.wait_stream(side)
    return [first] + rest


def cne_forward_many(mod, calls):
    """Run several independent CNE calls in lock-step: everything is per call except the Bi-LSTM recurrence, which is ONE
    launch over all streams of all calls (it is latency-bound by its longest sequence, not throughput-bound).  The per-call
    phases of the first (small) call run on a second HIP stream, filling the gaps of the big call's kernels."""
    H = mod.hidden_dim
    dev = (calls[0][0][0] if isinstance(calls[0][0], tuple) else calls[0][0]).device
    if len(calls) > 1:
        mod._packed_weights('title', mod.title_lstm)    # (re)pack on the main stream BEFORE forking: both calls read them
        mod._packed_weights('content', mod.content_lstm)
    else:
        # one call (the union of candidate and history call): the two re-packs need nothing of this step but the parameters, so they
        # run on the leaf stream next to the planner / row gather of the chains; each chain waits for them in front of its projection
        key = (dev.type, dev.index)
        if key not in ops._LEAF:
            ops._LEAF[key] = ops.new_stream(dev)
        leaf = ops._LEAF[key]
        leaf.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(leaf):
            mod._packed_weights('title', mod.title_lstm)
            mod._packed_weights('content', mod.content_lstm)
            mod.__dict__['_packed_ev'] = torch.cuda.Event()
            mod.__dict__['_packed_ev'].record()
    if mod.training:
        # the bf16 images of the parameters (and of their cached transposes) the gate / attention / user-encoder GEMMs will read: re-split on the LEAF
        # stream HERE -- behind the weight packing (which the input projection waits for: in front of it the splits cost +0.3 ms) and in FRONT of the
        # token sorts (needed by the backward pass only), i.e. under the input projection.  (Under the forward recurrence, whose workgroups fill every
        # CU, the 29 launches of ~5 us crawled at 70-160 us each and ended level with their first users: profiles/r06_ab.txt, first collection of the round.)
        ops.bx3_prefetch(dev)
    pre = _fork_join(len(calls), dev, lambda i, on_main: _cne_fwd_pre(mod, *calls[i], par=on_main))
    items = [st for sv in pre for st in (sv['streams'][1],)] + [st for sv in pre for st in (sv['streams'][0],) if not st.get('lstm_done')]   # content streams first
    for i in range(0, len(items), 4):
        ops.lstm_fwd(items[i:i + 4], H)
    return _fork_join(len(calls), dev, lambda i, on_main: _cne_fwd_post(mod, pre[i], on_main))


def _torch_tie_perm(masks):
    """tie_order='torch': the installed torch's CPU sort decides the order of equal lengths, per call (as the reference's
    torch.sort calls do); for a union of calls, any length-descending order that keeps each call's own order."""
    perms, lens = [], []
    base = 0
    for m in masks:
        l = (m.sum(dim=1).long() + (~m[:, 0]).long()).cpu()      # lengths after the mask[:,0]=1 fix
        q = torch.sort(l, descending=True)[1]
        perms.append(q + base)
        lens.append(l[q])
        base += m.shape[0]
    if len(perms) == 1:
        return perms[0].to(torch.int32)
    q, l = torch.cat(perms), torch.cat(lens)
    return q[torch.sort(l, descending=True, stable=True)[1]].to(torch.int32)


def _cne_fwd_pre(mod, title_text, title_mask, content_text, content_mask, category, subCategory, par=False, partner=None):
    """partner (optional, single call only): (pc_of_title, pt_of_content) int64 [n] -- for sequence i, the sequence whose CONTENT
    memory gates title i / whose TITLE memory gates content i -- instead of the call's own rank pairing (see cne_history_dedup)."""
    union = isinstance(title_text, tuple)            # (candidate call's tensor, history call's tensor) per argument
    if union:
        B, N = title_text[0].shape[:2]
        n0 = B * N
        n = n0 + B * title_text[1].shape[1]
        dev = title_text[0].device
    else:
        B, N = title_text.shape[:2]
        n = n0 = B * N
        dev = title_text.device
    H, E, A = mod.hidden_dim, mod.word_embedding_dim, mod.attention_dim
    H2 = 2 * H
    f32 = dict(device=dev, dtype=torch.float32)
    p = mod.dropout_rate if mod.training else 0.0
    seed = mod._next_seed()
    emb = mod.word_embedding.weight

    streams = [None, None]
    proj_done = {}

    def prepare(slot, name, ids, mask, Lx, lstm, Hlin, Mlin, satt, catt):
        parts = list(zip(ids, mask)) if union else [(ids, mask)]
        ids2 = [_i32(i).reshape(-1, Lx).contiguous() for i, _ in parts]
        mask2 = [m.view(-1, Lx) for _, m in parts]      # views: the in-place mask[:,0]=1 must reach the caller's tensors
        perm = _torch_tie_perm(mask2).to(dev) if mod.tie_order == 'torch' else None
        plan = ops.SeqPlan(mask2[0], ids2[0], perm, *((mask2[1], ids2[1]) if union else ()))
        plan_ev = None
        if union and name == 'content':
            plan_ev = torch.cuda.Event()
            plan_ev.record()
        w = mod._packed_weights(name, lstm)
        cap = plan.cap
        pev = mod.__dict__.get('_packed_ev')
        if pev is not None:
            torch.cuda.current_stream(dev).wait_event(pev)
        st = dict(name=name, L=Lx, plan=plan, plan_ev=plan_ev, w=w, lstm=lstm, Hlin=Hlin, Mlin=Mlin, satt=satt, catt=catt, seed=seed + _SITE[name])
        st['gates'] = torch.empty((cap, 2 * w.NP), **f32)
        # dropout(embedding rows) materialised ONCE per token (6 TB/s gather): fused into the GEMM's A loader the counter hash
        # is recomputed by each of the 21 column blocks (-16 % on this GEMM and on the dW_ih GEMM of the backward)
        st['xd'] = ops.embed_gather(emb, plan.tok, p, st['seed'], dyn=plan.total)
        hook = mod.__dict__.get('_tokens_hook')
        if hook is not None and (mod.training or torch.is_grad_enabled()):
            # data parallel: the touched-row exchange of the table gradient learns this stream's words -- whenever a backward pass may
            # follow (also an eval-mode gradient check: the backward's table hook is unconditional; round-5 advisor)
            hook(plan.tok, plan.total)
        if ops.SCATTER_SORTED and mod.training and E <= 320:
            # the backward's embedding-row gradient is a segmented reduction over the rows sorted by word id (reproducible, no atomic
            # ceiling): the sort needs only the planned ids and runs on the leaf stream under the forward pass
            st['tsort'] = ops.TokenSort(plan.tok, plan.total, emb.shape[0])
        if _PROJ_ORDER and par and name == 'content' and 'ev' in proj_done:
            torch.cuda.current_stream(dev).wait_event(proj_done['ev'])      # A/B: content projection BEHIND the title projection (see _PROJ_ORDER)
        ops.gemm(st['xd'], w.w_ihp, st['gates'], M=cap, N=2 * w.NP, K=E, lda=E, ldb=E, ldc=2 * w.NP, dyn=plan.total, dyn_dim=1, bias=w.b_p,
                 flop_scale=4.0 * H / w.NP, tile=_PROJ_TILE)
        if _PROJ_ORDER and par and name == 'title':
            proj_done['ev'] = torch.cuda.Event()
            proj_done['ev'].record()
        st['cell'] = torch.empty((cap, 2 * w.HP), **f32)
        st['hout'] = torch.empty((cap, H2), **f32)
        st['cn'] = torch.empty((n, H2), **f32)
        if _LSTM_FWD_SPLIT and par and name == 'title':
            # the title recurrence (32 steps, a fifth of the tokens) right behind its own projection on the title stream: it runs under
            # the content stream's projection GEMM instead of adding to the recurrence launch the whole forward pass waits for
            ops.lstm_fwd([st], H)
            st['lstm_done'] = True
        streams[slot] = st

    # the title and the content stream are independent up to the recurrence: plan + gather + input projection of the
    # (small) title stream run next to the content stream's on the big call
    _two_chains(dev, par,
                lambda: prepare(0, 'title', title_text, title_mask, mod.max_title_length, mod.title_lstm, mod.title_H, mod.title_M,
                                mod.title_self_attention, mod.title_cross_attention),
                lambda: prepare(1, 'content', content_text, content_mask, mod.max_content_length, mod.content_lstm, mod.content_H,
                                mod.content_M, mod.content_self_attention, mod.content_cross_attention))
    sv = dict(streams=streams, n=n, n0=n0, union=union, B=B, N=N, p=p, seed=seed, category=category, subCategory=subCategory)
    if union:
        # rank pairing (newsEncoders.py:128-129): inside ONE call the partner of sorted position r is the other stream's position r;
        # in a union of two calls it is the other stream's position with the same (call, position inside the call).  Needs both
        # plans and nothing else: issued on the title stream behind the title projection, it is done long before the recurrence ends
        s2 = _title_stream(dev)
        with torch.cuda.stream(s2):
            s2.wait_event(streams[1]['plan_ev'])
            sv['pm'] = ops.cne_pair_map(streams[0]['plan'], streams[1]['plan'])
        sv['pm_stream'] = s2
    if partner is not None:
        assert not union
        pc_t, pt_c = partner
        pt, pc = streams[0]['plan'], streams[1]['plan']
        # title at sorted position s is sequence order_t[s]; its partner content is sequence pc_t[.], at content position rank_c[.]
        sv['pm_override'] = (pc.rank.long()[pc_t[pt.order.long()]].to(torch.int32).contiguous(),
                             pt.rank.long()[pt_c[pc.order.long()]].to(torch.int32).contiguous())
    return sv


def _cne_fwd_post(mod, sv, par=False):
    t_, c_ = sv['streams']
    n, B, N, p, seed = sv['n'], sv['B'], sv['N'], sv['p'], sv['seed']
    dev = t_['gates'].device
    H, E, A = mod.hidden_dim, mod.word_embedding_dim, mod.attention_dim
    H2 = 2 * H
    D = mod.news_embedding_dim
    f32 = dict(device=dev, dtype=torch.float32)

    if sv['union']:
        torch.cuda.current_stream(dev).wait_stream(sv['pm_stream'])
    t_['pm'], c_['pm'] = sv['pm'] if sv['union'] else sv.get('pm_override', (None, None))

    def gate_and_self(st, other):
        plan, cap = st['plan'], st['plan'].cap
        # title_M(sorted_content_m): both indexed by sorted RANK (newsEncoders.py:128-129)
        if st['pm'] is not None and _CN_SPLIT:
            # rank pairing over a union of two calls: the plain (skinny, 16 x 80 tiles) projection of every c_n row, then a row gather
            # through the pairing map -- the row-gathering 64 x 80 GEMM took 50-75 us here and 150 us in the backward pass (right in
            # front of the backward recurrence, beside the leaf stream's weight-gradient GEMMs)
            st['mproj'] = ops.embed_gather(ops.linear_fwd(other['cn'], st['Mlin'].weight, st['Mlin'].bias), st['pm'], 0.0, 0)
        else:
            st['mproj'] = ops.linear_fwd(other['cn'], st['Mlin'].weight, st['Mlin'].bias, **({} if st['pm'] is None else {'a_idx': st['pm']}))
        st['G'] = torch.empty((cap, H2), **f32)
        st['Ht'] = torch.empty((cap, H2), **f32)
        ops.gemm(st['hout'], st['Hlin'].weight, st['Ht'], M=cap, N=H2, K=H2, lda=H2, ldb=H2, ldc=H2, dyn=plan.total, dyn_dim=1,
                 rowvec=st['mproj'], ldrv=H2, rowvec_map=plan.row_seq, act=ops.ACT_SIGMOID, aux_out=st['G'], ldaux=H2,
                 mul=st['hout'], ldmul=H2, tile=_GATE_TILE)
        st['th'] = torch.empty((cap, A), **f32)
        sa = st['satt']
        ops.gemm(st['Ht'], sa.affine1.weight, st['th'], M=cap, N=A, K=H2, lda=H2, ldb=H2, ldc=A, dyn=plan.total, dyn_dim=1,
                 bias=sa.affine1.bias, act=ops.ACT_TANH, tile=_GATE_TILE)
        st['alpha_s'] = torch.empty(cap, **f32)
        st['selfv'] = torch.empty((n, H2), **f32)
        if A <= 256 and A % 4 == 0:
            # the w2 . tanh(.) score is computed inside the pool's own pass over the tokens (one launch less on the chain)
            ops.pool_fwd(x=st['Ht'], ldx=H2, D=H2, n=n, Lx=st['L'], plan=plan, th=st['th'], w2=sa.affine2.weight, alpha=st['alpha_s'],
                         out=st['selfv'], ldo=H2)
        else:
            score = torch.empty(cap, **f32)
            ops.rowdot(st['th'], sa.affine2.weight, score, dyn=plan.total)
            ops.pool_fwd(x=st['Ht'], ldx=H2, D=H2, n=n, Lx=st['L'], plan=plan, score=score, alpha=st['alpha_s'], out=st['selfv'], ldo=H2)

    _two_chains(dev, par, lambda: gate_and_self(t_, c_), lambda: gate_and_self(c_, t_))

    rep = torch.empty((n, D), **f32)

    def cross(st, other, col0):
        plan, ca = st['plan'], st['catt']
        st['qv'] = ops.linear_fwd(other['selfv'], ca.Q.weight, ca.Q.bias)                    # [n, A]
        st['v'] = torch.empty((n, H2), **f32)
        ops.gemm(st['qv'], ca.K.weight, st['v'], M=n, N=H2, K=A, lda=A, ldb=H2, ldc=H2, trans_b=True)   # K^T (Q q + b)
        st['alpha_c'] = torch.empty(plan.cap, **f32)
        ops.pool_fwd(x=st['Ht'], ldx=H2, D=H2, n=n, Lx=st['L'], plan=plan, v=st['v'], ldv=H2, scale=1.0 / math.sqrt(A),
                     alpha=st['alpha_c'], out=rep[:, col0:], ldo=D, add_in=st['selfv'], ldadd=H2)

    _two_chains(dev, par, lambda: cross(t_, c_, 0), lambda: cross(c_, t_, H2))
    # feature fusion (newsEncoders.py:50-54): category + subCategory rows of both calls in ONE launch, the two calls' id tensors read
    # where they lie (round 2: torch.cat x 2 + two launches)
    if sv['union']:
        cat0, cat1 = (_i32(x).reshape(-1).contiguous() for x in sv.pop('category'))
        sub0, sub1 = (_i32(x).reshape(-1).contiguous() for x in sv.pop('subCategory'))
    else:
        cat0, cat1 = _i32(sv.pop('category')).reshape(n).contiguous(), None
        sub0, sub1 = _i32(sv.pop('subCategory')).reshape(n).contiguous(), None
    cd, sd = mod.category_embedding.weight.shape[1], mod.subCategory_embedding.weight.shape[1]
    ops.fusion_rows_fwd(mod.category_embedding.weight, mod.subCategory_embedding.weight, cat0, sub0, cat1, sub1, rep[:, 2 * H2:], D, p,
                        seed + _SITE['cat'], seed + _SITE['sub'])
    sv.update(cats=(cat0, sub0, cat1, sub1), cd=cd, sd=sd)
    for st in sv['streams']:                             # not needed by backward
        st.pop('mproj')
    if sv['union']:
        n0 = sv['n0']
        return (rep[:n0].view(B, N, D), rep[n0:].view(B, (n - n0) // B, D)), sv
    return rep.view(B, N, D), sv


def cne_backward(mod, sv, drep):
    cne_backward_many(mod, [(sv, drep)])


def cne_backward_many(mod, pairs):
    """Backward of cne_forward_many: per-call stages (call 0 on the side stream) around ONE shared recurrence-backward launch."""
    H = mod.hidden_dim
    dev = pairs[0][1].device
    for q in mod.parameters():          # materialise (zero-fill) missing .grad buffers on the main stream BEFORE forking
        grad_of(q)
    mod.__dict__['_table_scatters'] = 2 * len(pairs)      # embedding-row scatter GEMMs of this pass (title + content per call)
    with ops.leaf_scope(dev) as leaf:        # weight-gradient GEMMs of the pre phase: leaves, joined after the recurrence + post phase
        _cne_bwd_rest(mod, pairs, H, dev, leaf)


def _cne_bwd_rest(mod, pairs, H, dev, leaf):
    _fork_join(len(pairs), dev, lambda i, on_main: _cne_bwd_pre(mod, pairs[i][0], pairs[i][1], on_main, leaf))

    # recurrence backward: ONE launch over every token stream (longest tiles of all streams first, lstm.hip pair_id), then the
    # token-reduction GEMMs per stream kind -- the title streams' on the side stream next to the content streams'.  All
    # parameter-gradient accumulation below is atomic.  NNR_LSTM_BWD_SPLIT=1: one launch per stream kind on two HIP streams
    # (round 1's layout: the title recurrence then queues behind the content stream's GEMMs for CUs).
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    if _BWD_SPLIT:
        def run_kind(kind):
            ops.lstm_bwd([sv['streams'][kind] for sv, _ in pairs], H)
            for sv, _ in pairs:
                _cne_bwd_post(mod, sv, sv['streams'][kind], leaf if kind == 1 else None)

        side.wait_stream(main)
        with torch.cuda.stream(side):
            run_kind(0)
        run_kind(1)
        main.wait_stream(side)
        return
    ops.lstm_bwd([sv['streams'][1] for sv, _ in pairs] + [sv['streams'][0] for sv, _ in pairs], H)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for sv, _ in pairs:
            _cne_bwd_post(mod, sv, sv['streams'][0], None)
    for sv, _ in pairs:
        _cne_bwd_post(mod, sv, sv['streams'][1], leaf)
    main.wait_stream(side)


def _cne_bwd_pre(mod, sv, drep, par=False, leaf=None):
    if leaf is None:
        leaf = lambda fn, *tensors, **kw: fn()
    t_, c_ = sv['streams']
    n, p, seed = sv['n'], sv['p'], sv['seed']
    H, E, A = mod.hidden_dim, mod.word_embedding_dim, mod.attention_dim
    H2 = 2 * H
    D = mod.news_embedding_dim
    dev = drep.device
    f32 = dict(device=dev, dtype=torch.float32)
    drep = drep.reshape(n, D)
    emb = mod.word_embedding.weight

    # (a leaf of the backward pass: nothing downstream reads the two table gradients -- on the leaf stream, off the dependent chain)
    leaf(lambda: ops.fusion_rows_bwd(*sv['cats'], sv['cd'], sv['sd'], drep[:, 2 * H2:], D, grad_of(mod.category_embedding.weight),
                                     grad_of(mod.subCategory_embedding.weight), p, seed + _SITE['cat'], seed + _SITE['sub']), drep)

    # the title stream's weight-gradient GEMMs on a SECOND leaf stream when the step is small and latency-bound (content stream <= 64 k token
    # rows = per-GPU batch 8: the leaf stream was the last to finish, 3.27-3.31 -> 3.23-3.24 ms); at batch 16 neutral, at batch 64 -- where the
    # leaf work is throughput-bound -- 0.1 ms SLOWER (10.72-10.77 vs 10.61-10.65), hence the bound
    def alt_leaf(st):
        return st['name'] == 'title' and c_['plan'].cap <= _LEAF2_ROWS

    # ---- cross attention pools: dv -> K / Q params and the gradient of the OTHER stream's self vector.  Round 5 (_POOL_FUSED): this pass no
    # longer writes dHt -- it leaves its per-token d score (ds_c) behind, and the self pool's backward below writes
    # dHt = alpha_s d self + alpha_c d cross + scale ds_c v in ONE store (before: store here, read-modify-write there: 5 passes over
    # [tokens, 400] for the two pools instead of 3; the cross pool must still run first, its dv feeds the other stream's self vector)
    def cross_bwd(st, other, col0):
        plan, ca, cap = st['plan'], st['catt'], st['plan'].cap
        dv = torch.empty((n, H2), **f32)
        if _POOL_FUSED:
            st['ds_c'] = torch.empty(cap, **f32)
            ops.pool_bwd(x=st['Ht'], ldx=H2, D=H2, n=n, Lx=st['L'], plan=plan, v=st['v'], ldv=H2, scale=1.0 / math.sqrt(A),
                         alpha=st['alpha_c'], dout=drep[:, col0:], lddo=D, dscore=st['ds_c'], dv=dv, lddv=H2)
        else:
            st['dHt'] = torch.empty((cap, H2), **f32)
            ops.pool_bwd(x=st['Ht'], ldx=H2, D=H2, n=n, Lx=st['L'], plan=plan, v=st['v'], ldv=H2, scale=1.0 / math.sqrt(A),
                         alpha=st['alpha_c'], dout=drep[:, col0:], lddo=D, dx=st['dHt'], lddx=H2, dv=dv, lddv=H2)
        dqv = torch.empty((n, A), **f32)
        ops.gemm(dv, ca.K.weight, dqv, M=n, N=A, K=H2, lda=H2, ldb=H2, ldc=A)                 # dqv = dv . K^T
        leaf(lambda: (ops.linear_bwd_weight(st['qv'], dv, grad_of(ca.K.weight)),             # dK[A,H2] += qv^T dv
                      ops.linear_bwd_weight(dqv, other['selfv'], grad_of(ca.Q.weight), db=grad_of(ca.Q.bias))), dv, dqv, alt=alt_leaf(st))      # (bias gradient fused: column sums of dqv)
        other['dself_x'] = ops.linear_bwd_data(dqv, ca.Q.weight)                              # grad of other.selfv via the query

    _two_chains(dev, par, lambda: cross_bwd(t_, c_, 0), lambda: cross_bwd(c_, t_, H2))

    # ---- self attention pools, additive score, gate
    def self_gate_bwd(st, other, col0):
        plan, sa, cap = st['plan'], st['satt'], st['plan'].cap
        ds = torch.empty(cap, **f32)
        if _POOL_FUSED:
            st['dHt'] = torch.empty((cap, H2), **f32)
            ops.pool_bwd(x=st['Ht'], ldx=H2, D=H2, n=n, Lx=st['L'], plan=plan, score=None, alpha=st['alpha_s'],
                         dout=drep[:, col0:], lddo=D, dout2=st['dself_x'], lddo2=H2, dx=st['dHt'], lddx=H2, dscore=ds,
                         alpha_b=st['alpha_c'], dout_b=drep[:, col0:], lddo_b=D, dscore_b=st['ds_c'], v_b=st['v'], ldv_b=H2, scale_b=1.0 / math.sqrt(A))
            st['ds_c'] = None
        else:
            ops.pool_bwd(x=st['Ht'], ldx=H2, D=H2, n=n, Lx=st['L'], plan=plan, score=None, alpha=st['alpha_s'],
                         dout=drep[:, col0:], lddo=D, dout2=st['dself_x'], lddo2=H2, dx=st['dHt'], lddx=H2, dx_accumulate=True, dscore=ds)
        th = st['th']
        ops.tanh_score_bwd(th, ds, sa.affine2.weight, grad_of(sa.affine2.weight), plan, A)    # th := dpre
        st['dH'] = torch.empty((cap, H2), **f32)
        dpre = torch.empty((cap, H2), **f32)             # (not Ht's buffer: the deferred weight-gradient GEMM below still reads it)
        if _GATE_FUSED:
            # the GEMM that completes dHt (+= d tanh-projection . W1) applies the gate's backward in its epilogue: Ht = hout * G ->
            # dH = dHt * G, d pre = dHt * hout * G * (1 - G) -- one launch and one pass over dHt less on the dependent chain (round 4)
            ops.gemm(th, ops.wt(sa.affine1.weight), st['dH'], M=cap, N=H2, K=A, lda=A, ldb=A, ldc=H2, dyn=plan.total, dyn_dim=1,
                     pre_add=st['dHt'], ldpre=H2, gate_bwd=True, mul=st['G'], ldmul=H2, resid=st['hout'], ldres=H2, aux_out=dpre, ldaux=H2,
                     tile=_GATE_TILE)
        else:
            ops.gemm(th, ops.wt(sa.affine1.weight), st['dHt'], M=cap, N=H2, K=A, lda=A, ldb=A, ldc=H2, accumulate=True,      # NT on W1^T
                     dyn=plan.total, dyn_dim=1)
            ops.gate_bwd(st['dHt'], st['hout'], st['G'], st['dH'], dpre, plan, H2)               # gate: Ht = hout * G
        leaf(lambda: ops.linear_bwd_weight(th, st['Ht'], grad_of(sa.affine1.weight), dyn=plan.total, db=grad_of(sa.affine1.bias)), th, st['Ht'], alt=alt_leaf(st))
        ops.gemm(dpre, ops.wt(st['Hlin'].weight), st['dH'], M=cap, N=H2, K=H2, lda=H2, ldb=H2, ldc=H2, accumulate=True,   # NT on W_H^T
                 dyn=plan.total, dyn_dim=1, tile=_GATE_TILE)
        leaf(lambda: ops.linear_bwd_weight(dpre, st['hout'], grad_of(st['Hlin'].weight), dyn=plan.total), dpre, st['hout'], alt=alt_leaf(st))
        dP = torch.empty((n, H2), **f32)                  # d mproj[rank]
        ops.packed_seq_sum(dpre, H2, plan, dP)
        pm, opm = st['pm'], other['pm']                   # union of two calls: mproj[s] = M(cn_other[pm[s]])  (pm^-1 = other's pm)
        leaf(lambda: ops.linear_bwd_weight(dP, other['cn'], grad_of(st['Mlin'].weight), db=grad_of(st['Mlin'].bias),
                                           **({} if pm is None else {'b_idx': pm})), dP, alt=alt_leaf(st))
        # (NNR_DCN_TILE: this [n, 400] x [400, 400] product runs beside the leaf stream's weight-gradient GEMMs, whose workgroups hold 120 KB of a
        #  CU's LDS, and its 74 KB tile waits for them: 277 us in the step against 40 us alone.  The 19 KB register-staged tile (2) halves its
        #  in-step time and leaves the step where it was -- the chain waits for the same leaf work one call later; profiles/r06_ab.txt call 24)
        dcn_kw = {'tile': _DCN_TILE} if _DCN_TILE else {}
        if opm is not None and _CN_SPLIT:
            other['dcn'] = ops.embed_gather(ops.linear_bwd_data(dP, st['Mlin'].weight, **dcn_kw), opm, 0.0, 0)            # [n, H2], rank-indexed
        else:
            other['dcn'] = ops.linear_bwd_data(dP, st['Mlin'].weight, **dcn_kw, **({} if opm is None else {'a_idx': opm}))   # [n, H2], rank-indexed
        st['dHt'] = None

    _two_chains(dev, par, lambda: self_gate_bwd(t_, c_, 0), lambda: self_gate_bwd(c_, t_, H2))

    for st in (t_, c_):
        st['dh'] = st['dH']


def _cne_bwd_post(mod, sv, st, leaf=None):
    """After the recurrence backward of token stream `st`: weight / bias gradients of the LSTM and the embedding-row scatter.
    All of it is leaf work; with `leaf` given the three weight-gradient GEMMs go to the leaf stream and run next to the scatter
    GEMM (each has a partially filled last wave of workgroups that the other fills)."""
    p = sv['p']
    H, E = mod.hidden_dim, mod.word_embedding_dim
    H2 = 2 * H
    f32 = dict(device=st['gates'].device, dtype=torch.float32)
    emb = mod.word_embedding.weight
    plan, w, cap = st['plan'], st['w'], st['plan'].cap
    dg = st['gates']                                  # now d(pre-activation gates), p-order
    NP = w.NP

    # packed LSTM weight-gradient accumulators: one persistent, zeroed workspace per (token stream, encoder call); the unpack
    # kernel hands it back zeroed (three fill launches per stream and step otherwise -- 0.5 ms of kernel time in round 1)
    ws = mod.__dict__.setdefault('_dw_ws', {})
    key = (st['name'], sv['n'])
    if key not in ws or ws[key][0].shape != (2 * NP, E) or ws[key][0].device != dg.device:
        ws[key] = (torch.zeros((2 * NP, E), **f32), torch.zeros(2 * NP, **f32), torch.zeros((2, NP, H), **f32))
    dw_ihp, db_p, dw_hhp = ws[key]
    ops.tape_keep(dw_ihp, db_p, dw_hhp)

    def dw_ih():
        t, bm, bn, target = ops.tn_tile(2 * NP, E, cap)
        ops.gemm(dg, st['xd'], dw_ihp, M=2 * NP, N=E, K=cap, lda=2 * NP, ldb=E, ldc=E, trans_a=True, trans_b=True,
                 split_k=ops.split_for(2 * NP, E, cap, bm, bn, target), atomic=True, dyn=plan.total, dyn_dim=2, colsum_out=db_p, tile=t,
                 flop_scale=4.0 * H / NP)

    def dw_hh(d):
        t, bm, bn, target = ops.tn_tile(NP, H, cap, gather=True)
        ops.gemm(dg[:, d * NP:], st['hout'][:, d * H:], dw_hhp[d], M=NP, N=H, K=cap, lda=2 * NP, ldb=H2, ldc=H, trans_a=True,
                 trans_b=True, b_idx=(plan.prev_f, plan.prev_r)[d], split_k=ops.split_for(NP, H, cap, bm, bn, target), atomic=True,
                 dyn=plan.total, dyn_dim=2, tile=t, flop_scale=4.0 * H / NP)

    def table_hook():
        hook = mod.__dict__.get('_table_scatter_hook')       # data parallel: the table's gradient bucket goes out after the last scatter
        if hook is not None:
            hook(mod.__dict__.get('_table_scatters', 2))

    def dx_scatter():
        # d(embedding rows): dX = dgates . W_ihp (NT on the transposed packed weight), scattered (atomic) into the table
        # gradient through the dropout mask.  The title streams' launch runs BESIDE the content recurrence, whose workgroups
        # hold 98 KB of LDS per CU: there the 40 KB tile (tile 15) can move in next to them, the 80 KB one (the automatic
        # choice for this long reduction) cannot and crawls (384 us for 4 GFLOP, measured).
        if _DX_SPLIT:
            # plain-store GEMM into the cell-state buffer (dead after the recurrence backward; xd / hout are still being read by the
            # weight-gradient GEMMs on the leaf stream) + the row-scatter kernel: the atomic epilogue of the fused form costs the GEMM
            # 18 % (764 vs 624 us alone, tools/dx_epilogue_bench.py)
            # (the cell buffer is [cap, 2*HP]: large enough only when 2*HP >= E -- not at --hidden_dim <= 144 with E = 300)
            dx = st['cell'].view(-1)[:cap * E].view(cap, E) if st['cell'].numel() >= cap * E else torch.empty((cap, E), **f32)
            ops.gemm(dg, w.w_ihp_t, dx, M=cap, N=E, K=2 * NP, lda=2 * NP, ldb=2 * NP, ldc=E, dyn=plan.total, dyn_dim=1,
                     tile=_DX_TILE if leaf is not None else _TITLE_DX_TILE, flop_scale=4.0 * H / NP)
            if st.get('tsort') is not None:
                ops.embed_scatter_sorted(dx, st['tsort'], grad_of(emb), p, st['seed'])
            else:
                ops.embed_scatter(dx, plan.tok, grad_of(emb), p, st['seed'], dyn=plan.total)
            return
        ops.gemm(dg, w.w_ihp_t, grad_of(emb), M=cap, N=E, K=2 * NP, lda=2 * NP, ldb=2 * NP, ldc=E, c_idx=plan.tok, atomic=True,
                 drop=(4, p, st['seed'], E), dyn=plan.total, dyn_dim=1, tile=0 if leaf is not None else _TITLE_DX_TILE, flop_scale=4.0 * H / NP)

    # Data parallelism (world > 1), NNR_DP_TABLE_FIRST (opt-in: its benefit needs >= 2 GPUs to measure): the word-embedding table is 70 % of the
    # gradient bytes and its bucket can only leave after the LAST embedding-row scatter.  Both token streams then compute their
    # embedding-row gradient FIRST -- the weight-gradient GEMMs of the stream are issued behind the scatter instead of beside the dX
    # GEMM --, so the table bucket's all-reduce (72 MB: ~0.8 ms on one xGMI ring) starts ~2 ms before the end of the backward pass
    # and is hidden behind those GEMMs, instead of ~0.5 ms before it.  Single-GPU cost of this order: +0.05-0.1 ms.
    table_first = _DP_TABLE_FIRST == 1 or (_DP_TABLE_FIRST < 0 and mod.__dict__.get('_table_scatter_hook') is not None and _dp_world() > 1)
    if table_first:
        dx_scatter()
        table_hook()
        if leaf is None:
            dw_ih(); dw_hh(0); dw_hh(1)
        else:
            leaf(lambda: (dw_ih(), dw_hh(1)), dw_ihp, db_p, dw_hhp)         # (the leaf stream waits for this stream: behind the scatter)
            dw_hh(0)
            leaf.sync()
        ops.lstm_unpack_grads(dw_ihp, db_p, dw_hhp, H, E, [grad_of(q) for q in st['lstm'].param_list()], zero_src=True)
        return
    if leaf is None:
        # title streams (side stream, beside the content recurrence): the scatter GEMM first -- it is the largest launch and the
        # one the tail of the step would otherwise still be waiting for
        if _TITLE_DX_FIRST:
            dx_scatter()
            table_hook()
        dw_ih(); dw_hh(0); dw_hh(1)
    else:
        # two balanced halves: leaf stream dW_ih + dW_hh(reverse), this stream the scatter GEMM + dW_hh(forward)
        if _POST_INLINE:
            dx_scatter()
            table_hook()
            dw_ih(); dw_hh(0); dw_hh(1)
        else:
            leaf(lambda: (dw_ih(), dw_hh(1)), dw_ihp, db_p, dw_hhp)
            if _DWHH_FIRST:
                dw_hh(0)
            dx_scatter()
            table_hook()
            if not _DWHH_FIRST:
                dw_hh(0)
            leaf.sync()
    ops.lstm_unpack_grads(dw_ihp, db_p, dw_hhp, H, E, [grad_of(q) for q in st['lstm'].param_list()], zero_src=True)
    if leaf is None and not _TITLE_DX_FIRST:
        dx_scatter()
        table_hook()


@torch.no_grad()
def cne_history_dedup(mod, title_text, title_mask, content_text, content_mask, category, subCategory):
    """SURVEY.md section 8 f-3, the EXACT part for CNE (inference, dropout off): every short history is right-padded with news 0
    (MIND_corpus.py:352-353,369) -- on MIND-shaped batches half of the 50 slots per user -- and userEncoders.py:76-78 runs the news
    encoder over all of them.  A PAD slot's representation depends only on its own (constant) inputs and on the two cell states that
    gate it (newsEncoders.py:128-129: the content at its title's sorted rank, the title at its content's sorted rank).  When both of
    those belong to PAD-like sequences too (one token, id 0 -- their cell states are one constant each), the slot's representation is
    THE constant r_pad.  Those slots are dropped from the encoder call, one representative computes r_pad, and every kept sequence
    keeps its original partner (a dropped partner is replaced by the representative: same cell state).  Decided on the device from
    the ranks; one host read of the kept count.  Returns (representations [B, H, D], number of sequences actually encoded)."""
    assert not mod.training or mod.dropout_rate == 0.0, 'de-duplication is exact only without dropout (one mask per slot otherwise)'
    B, Hn = title_text.shape[:2]
    n, T, Cx = B * Hn, mod.max_title_length, mod.max_content_length
    dev = title_text.device
    tm, cm = title_mask.view(n, T), content_mask.view(n, Cx)
    tm[:, 0] = 1                                        # in place on the caller's tensors, as every CNE call does (newsEncoders.py:108-109)
    cm[:, 0] = 1
    tt, ct = _i32(title_text).reshape(n, T), _i32(content_text).reshape(n, Cx)
    tlen, clen = tm.sum(dim=1), cm.sum(dim=1)
    order_t = torch.sort(tlen, descending=True, stable=True)[1]
    order_c = torch.sort(clen, descending=True, stable=True)[1]
    rank_t, rank_c = torch.empty_like(order_t), torch.empty_like(order_c)
    ar = torch.arange(n, device=dev)
    rank_t[order_t], rank_c[order_c] = ar, ar
    pc_t, pt_c = order_c[rank_t], order_t[rank_c]       # partner sequences under the call's own rank pairing
    cat, sub = _i32(category).reshape(n), _i32(subCategory).reshape(n)
    padlike = (tlen == 1) & (tt[:, 0] == 0) & (clen == 1) & (ct[:, 0] == 0) & (cat == 0) & (sub == 0)
    drop = padlike & padlike[pc_t] & padlike[pt_c]
    nd = int(drop.sum())
    if nd <= 1 or mod.tie_order != 'stable':
        rep, _ = cne_forward(mod, title_text, title_mask, content_text, content_mask, category, subCategory)
        return rep, n
    rep_idx = torch.nonzero(drop)[0, 0]                                  # the representative PAD slot
    keep = ~drop
    keep[rep_idx] = True
    kept = torch.nonzero(keep).flatten()                                 # original index of compact sequence k
    compact = torch.full((n,), -1, device=dev, dtype=torch.long)
    compact[kept] = torch.arange(kept.numel(), device=dev)
    rep_c = compact[rep_idx]
    remap = lambda partner: torch.where(compact[partner[kept]] >= 0, compact[partner[kept]], rep_c)
    sel = lambda x, w: x.reshape(n, w)[kept].unsqueeze(0).contiguous() if w else x.reshape(n)[kept].unsqueeze(0).contiguous()
    (out, _), = [(_cne_fwd_post(mod, sv, False)) for sv in [_cne_fwd_pre_and_lstm(mod, sel(tt, T), sel(tm, T), sel(ct, Cx), sel(cm, Cx),
                                                                                 sel(cat, 0), sel(sub, 0), (remap(pc_t), remap(pt_c)))]]
    D = out.shape[-1]
    full = out.view(-1, D)[rep_c].expand(n, D).clone()
    full[kept] = out.view(-1, D)
    return full.view(B, Hn, D), int(kept.numel())


def _cne_fwd_pre_and_lstm(mod, tt, tm, ct, cm, cat, sub, partner):
    sv = _cne_fwd_pre(mod, tt, tm, ct, cm, cat, sub, par=False, partner=partner)
    ops.lstm_fwd([sv['streams'][1], sv['streams'][0]], mod.hidden_dim)
    return sv


class CNE(NewsEncoder):
    """newsEncoders.py:57-141."""

    def __init__(self, config, word_table=None):
        super().__init__(config, word_table)
        self.max_title_length = config.max_title_length
        self.max_content_length = config.max_abstract_length
        self.hidden_dim = config.hidden_dim
        self.attention_dim = config.attention_dim
        self.news_embedding_dim = config.hidden_dim * 4 + config.category_embedding_dim + config.subCategory_embedding_dim
        self.tie_order = getattr(config, 'tie_order', 'stable')
        h2 = self.hidden_dim * 2
        self.title_lstm = LSTMParams(self.word_embedding_dim, self.hidden_dim)
        self.content_lstm = LSTMParams(self.word_embedding_dim, self.hidden_dim)
        self.title_H = nn.Linear(h2, h2, bias=False)
        self.title_M = nn.Linear(h2, h2, bias=True)
        self.content_H = nn.Linear(h2, h2, bias=False)
        self.content_M = nn.Linear(h2, h2, bias=True)
        self.title_self_attention = Attention(h2, config.attention_dim)
        self.content_self_attention = Attention(h2, config.attention_dim)
        self.title_cross_attention = ScaledDotProduct_CandidateAttention(h2, h2, config.attention_dim)
        self.content_cross_attention = ScaledDotProduct_CandidateAttention(h2, h2, config.attention_dim)

    def initialize(self):
        super().initialize()
        for lstm in (self.title_lstm, self.content_lstm):
            for parameter in lstm.parameters():
                if len(parameter.size()) >= 2:
                    nn.init.orthogonal_(parameter.data)
                else:
                    nn.init.zeros_(parameter.data)
        gain = nn.init.calculate_gain('sigmoid')
        for lin in (self.title_H, self.title_M, self.content_H, self.content_M):
            nn.init.xavier_uniform_(lin.weight, gain=gain)
        nn.init.zeros_(self.title_M.bias)
        nn.init.zeros_(self.content_M.bias)
        self.title_self_attention.initialize()
        self.content_self_attention.initialize()
        self.title_cross_attention.initialize()
        self.content_cross_attention.initialize()

    def _packed_weights(self, name, lstm):
        """nn.LSTM parameters in the recurrent kernels' layouts, re-packed when the parameters change (once per optimizer
        step: both encoder calls of a step share them)."""
        cache = self.__dict__.setdefault('_pack_cache', {})
        key = (PARAM_EPOCH[0],) + tuple((q.data_ptr(), q._version) for q in lstm.param_list())
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            hit = (key, ops.LstmPacked(lstm.param_list(), self.hidden_dim, self.word_embedding_dim))
            cache[name] = hit
        return hit[1]

    def forward(self, title_text, title_mask, title_entity, content_text, content_mask, content_entity, category, subCategory, user_embedding):
        # title_entity / content_entity / user_embedding are accepted and ignored, as in the reference (newsEncoders.py:102-141)
        return _CNEFunction.apply(self.word_embedding.weight, self, title_text, title_mask, content_text, content_mask, category, subCategory)

    def forward_pair(self, cand, hist):
        """(candidate call, history call) -> (candidate reps, history reps); same results as two forward() calls, with the
        two calls' Bi-LSTM recurrences sharing one launch.  cand / hist = (title_text, title_mask, content_text, content_mask,
        category, subCategory)."""
        return _CNEPairFunction.apply(self.word_embedding.weight, self, *cand, *hist)


# ================================================================================================== MHSA / CNN
_MHSA_PACKED = os.environ.get('NNR_MHSA_PACKED', '1') != '0'      # A/B (round 5): MHSA news encoder over packed token rows


def mhsa_packed(enc, title_text):
    """Packed rows need the 4-head cooperative attention core (heads % 4 == 0, head_dim % 4 == 0, titles of at most 32 positions) and
    device tensors; anything else takes the dense path."""
    return (_MHSA_PACKED and title_text.is_cuda and enc.head_num % 4 == 0 and enc.head_dim % 4 == 0 and enc.head_dim <= 32
            and enc.max_sentence_length <= 32 and enc.word_embedding_dim % 4 == 0)


class MHSA(NewsEncoder):
    """newsEncoders.py:173-200: title only; embedding gather -> QKV GEMMs -> MFMA attention core -> dropout ->
    additive attention pool -> feature fusion."""
    batch_independent = True          # a news representation does not depend on the other news of the call (evaluate.py cache)

    def __init__(self, config, word_table=None):
        super().__init__(config, word_table)
        self.max_sentence_length = config.max_title_length
        self.head_num, self.head_dim = config.head_num, config.head_dim
        self.feature_dim = config.head_num * config.head_dim
        self.multiheadAttention = MultiHeadAttention(config.head_num, config.word_embedding_dim, config.max_title_length,
                                                     config.max_title_length, config.head_dim, config.head_dim)
        self.attention = Attention(config.head_num * config.head_dim, config.attention_dim)
        self.news_embedding_dim = config.head_num * config.head_dim + config.category_embedding_dim + config.subCategory_embedding_dim

    def initialize(self):
        super().initialize()
        self.multiheadAttention.initialize()
        self.attention.initialize()

    def forward(self, title_text, title_mask, title_entity, content_text, content_mask, content_entity, category, subCategory, user_embedding):
        from . import functional as Fn
        B, N = title_text.shape[:2]
        n, Lx = B * N, self.max_sentence_length
        p = self.dropout_rate if self.training else 0.0
        seed = self._next_seed()
        mask = title_mask.view(n, Lx)
        if mhsa_packed(self, title_text):
            # only the token rows that can reach the result (functional.MhsaPack): ~36 % of n * L at MIND title lengths
            pack = Fn.MhsaPack(mask, _i32(title_text).reshape(n, Lx))
            w = Fn.PackedEmbedDropFn.apply(self.word_embedding.weight, pack, p, seed + 1)                   # [cap, E], live rows only
            qkv = Fn.QKVFn.apply(w, self.multiheadAttention, pack.plan.total)
            c = Fn.PackedMhsaCoreFn.apply(qkv, mask, pack, self.head_num, self.head_dim, p, seed + 2)       # [cap, h*d], dropout fused
            rep = Fn.PackedAttentionFn.apply(c, self.attention, mask, pack)                                 # [n, h*d]
            return Fn.FuseFn.apply(rep, self, category, subCategory, p, seed).view(B, N, self.news_embedding_dim)
        w = Fn.EmbedDropFn.apply(self.word_embedding.weight, title_text, p, seed + 1)                       # [n*L, E]
        qkv = Fn.QKVFn.apply(w, self.multiheadAttention, None)
        c = Fn.MhsaCoreFn.apply(qkv, mask, n, Lx, self.head_num, self.head_dim, p, seed + 2)                # [n*L, h*d], dropout fused
        rep = self.attention(c.view(n, Lx, self.feature_dim), mask)                                         # [n, h*d]
        return Fn.FuseFn.apply(rep, self, category, subCategory, p, seed).view(B, N, self.news_embedding_dim)


class CNN(NewsEncoder):
    """newsEncoders.py:144-170: title only; embedding gather -> Conv1d(k=3)+ReLU as shifted GEMMs -> dropout_ ->
    additive attention pool -> feature fusion."""
    batch_independent = True

    def __init__(self, config, word_table=None):
        super().__init__(config, word_table)
        self.max_sentence_length = config.max_title_length
        self.cnn_kernel_num = config.cnn_kernel_num
        self.conv = Conv1D(config.cnn_method, config.word_embedding_dim, config.cnn_kernel_num, config.cnn_window_size)
        self.attention = Attention(config.cnn_kernel_num, config.attention_dim)
        self.news_embedding_dim = config.cnn_kernel_num + config.category_embedding_dim + config.subCategory_embedding_dim

    def initialize(self):
        super().initialize()
        self.attention.initialize()

    def forward(self, title_text, title_mask, title_entity, content_text, content_mask, content_entity, category, subCategory, user_embedding):
        from . import functional as Fn
        B, N = title_text.shape[:2]
        n, Lx = B * N, self.max_sentence_length
        p = self.dropout_rate if self.training else 0.0
        seed = self._next_seed()
        mask = title_mask.view(n, Lx)
        w = Fn.EmbedDropFn.apply(self.word_embedding.weight, title_text, p, seed + 1)
        c = Fn.Conv1dReluFn.apply(w, self.conv.conv, n, Lx)                                                # [n*L, C]
        c = Fn.DropoutFn.apply(c, p, seed + 2)
        rep = self.attention(c.view(n, Lx, self.cnn_kernel_num), mask)
        return Fn.FuseFn.apply(rep, self, category, subCategory, p, seed).view(B, N, self.news_embedding_dim)
