"""Autograd building blocks over the C-ABI for the MHSA / CNN / ATT encoders (dense [n, L, F] layouts).
Parameter gradients are accumulated straight into `param.grad` (see layers.grad_of); the Functions return gradients
only for activations."""
import math

import os
import torch

from . import ops
from .layers import grad_of


class EmbedDropFn(torch.autograd.Function):
    """dropout(word_embedding(ids))  -- nn.Embedding + in-place Dropout (newsEncoders.py:163,193)."""

    @staticmethod
    def forward(ctx, table, ids, p, seed):
        idx = ids.reshape(-1)
        idx = (idx if idx.dtype == torch.int32 else idx.to(torch.int32)).contiguous()
        out = ops.embed_gather(table, idx, p, seed)
        ctx.table, ctx.idx, ctx.p, ctx.seed = table, idx, p, seed
        return out

    @staticmethod
    def backward(ctx, dout):
        ops.embed_scatter(dout.contiguous(), ctx.idx, grad_of(ctx.table), ctx.p, ctx.seed)
        return None, None, None, None


# ---------------------------------------------------------------------------------------------- MHSA news encoder over PACKED token rows
# (round 5)  The reference's MHSA news encoder multiplies all n * L padded positions through W_Q / W_K / W_V and the additive attention's
# affine1 (newsEncoders.py:187-200, layers.py:134-136,168); ~64 % of those rows are padding whose keys are masked (-1e9) and whose pooled
# weight is exactly 0.  Here only the rows ops.mask_cover names exist (every position up to a title's last valid one; ALL positions of a
# fully masked title, whose softmax is uniform), packed time-major by nnr_seq_plan exactly like CNE's token streams; the attention core
# finds position t of title i through a row map, the pool through the plan.  The original mask still masks keys and pooled positions.
class MhsaPack:
    """Plan + row map of one encoder call over `mask` [n, L] (bool / uint8) and `ids` [n, L] int32."""

    def __init__(self, mask, ids):
        cover = ops.mask_cover(mask)
        self.plan = ops.SeqPlan(cover, ids.contiguous(), None)          # (its in-place mask[:, 0] = 1 acts on `cover`: a no-op there)
        self.rowmap = ops.seq_rowmap(self.plan)
        self.cover = cover
        # round 6: two titles of <= 16 positions per 32 x 32 attention problem (the core's matrix work does not depend on a title's length)
        self.pair = ops.mhsa_pair_map(self.plan, mask) if (ops.MHSA_PAIR and self.plan.L == 32) else None


class PackedEmbedDropFn(torch.autograd.Function):
    """dropout(word_embedding(ids)) over the packed rows only; mask index = packed row * E + column."""

    @staticmethod
    def forward(ctx, table, pack, p, seed):
        plan = pack.plan
        out = ops.embed_gather(table, plan.tok, p, seed, dyn=plan.total)
        ctx.table, ctx.plan, ctx.p, ctx.seed = table, plan, p, seed
        return out

    @staticmethod
    def backward(ctx, dout):
        plan = ctx.plan
        ops.embed_scatter(dout.contiguous(), plan.tok, grad_of(ctx.table), ctx.p, ctx.seed, dyn=plan.total)
        return None, None, None, None


class PackedMhsaCoreFn(torch.autograd.Function):
    """MhsaCoreFn over packed rows: qkv / out are [cap, .] with only the plan's live rows defined."""

    @staticmethod
    def forward(ctx, qkv, mask, pack, heads, dh, p=0.0, seed=0):
        qkv = qkv.contiguous()
        out = torch.empty((pack.plan.cap, heads * dh), device=qkv.device, dtype=torch.float32)
        paired = pack.pair is not None and heads % 4 == 0 and dh % 4 == 0 and 32 * dh <= 768
        if paired:
            ops.mhsa_fwd_paired(qkv, pack.pair, pack.plan, heads, dh, out, p, seed)
        else:
            ops.mhsa_fwd_packed(qkv, mask, pack.rowmap, pack.plan, heads, dh, out, p, seed)
        ctx.qkv, ctx.mask, ctx.pack, ctx.dims, ctx.drop, ctx.paired = qkv, mask, pack, (heads, dh), (p, seed), paired
        return out

    @staticmethod
    def backward(ctx, dout):
        heads, dh = ctx.dims
        dqkv = torch.empty_like(ctx.qkv)
        if ctx.paired:
            ops.mhsa_bwd_paired(ctx.qkv, ctx.pack.pair, ctx.pack.plan, dout.contiguous(), heads, dh, dqkv, *ctx.drop)
        else:
            ops.mhsa_bwd_packed(ctx.qkv, ctx.mask, ctx.pack.rowmap, ctx.pack.plan, dout.contiguous(), heads, dh, dqkv, *ctx.drop)
        return dqkv, None, None, None, None, None, None


class PackedAttentionFn(torch.autograd.Function):
    """layers.py:167-175 over packed rows [cap, F]: tanh GEMM on the live rows, then the packed softmax pool (w2 . tanh(.) score inside the
    pool's pass, the ORIGINAL [n, L] mask applied to the scores); out [n, F] in the caller's row order."""

    @staticmethod
    def forward(ctx, feature, mod, mask, pack):
        plan = pack.plan
        cap, F = feature.shape
        A = mod.affine1.weight.shape[0]
        x = feature.contiguous()
        f32 = dict(device=x.device, dtype=torch.float32)
        th = torch.empty((cap, A), **f32)
        ops.gemm(x, mod.affine1.weight, th, M=cap, N=A, K=F, lda=F, ldb=F, ldc=A, bias=mod.affine1.bias, act=ops.ACT_TANH, dyn=plan.total, dyn_dim=1)
        alpha = torch.empty(cap, **f32)
        out = torch.empty((plan.n, F), **f32)
        if A <= 256 and A % 4 == 0:
            ops.pool_fwd(x=x, ldx=F, D=F, n=plan.n, Lx=plan.L, plan=plan, mask=mask, th=th, w2=mod.affine2.weight, alpha=alpha, out=out, ldo=F)
        else:
            score = torch.empty(cap, **f32)
            ops.rowdot(th, mod.affine2.weight, score, dyn=plan.total)
            ops.pool_fwd(x=x, ldx=F, D=F, n=plan.n, Lx=plan.L, plan=plan, mask=mask, score=score, alpha=alpha, out=out, ldo=F)
        ctx.mod, ctx.mask, ctx.plan, ctx.saved = mod, mask, plan, (x, th, alpha, F, A)
        return out

    @staticmethod
    def backward(ctx, dout):
        mod, plan = ctx.mod, ctx.plan
        x, th, alpha, F, A = ctx.saved
        cap = plan.cap
        f32 = dict(device=x.device, dtype=torch.float32)
        dx = torch.empty((cap, F), **f32)
        ds = torch.empty(cap, **f32)
        ops.pool_bwd(x=x, ldx=F, D=F, n=plan.n, Lx=plan.L, plan=plan, mask=ctx.mask, alpha=alpha, dout=dout.contiguous(), lddo=F, dx=dx, lddx=F, dscore=ds)
        ops.tanh_score_bwd(th, ds, mod.affine2.weight, grad_of(mod.affine2.weight), plan, A)          # th := d(pre-activation)
        gw, gb = grad_of(mod.affine1.weight), grad_of(mod.affine1.bias)
        ops.leaf_deferred(x.device, cap, lambda: ops.linear_bwd_weight(th, x, gw, db=gb, dyn=plan.total), th, x)
        if ops.USE_WT and cap >= 1024 and (A & 3) == 0:
            ops.gemm(th, ops.wt(mod.affine1.weight), dx, M=cap, N=F, K=A, lda=A, ldb=A, ldc=F, accumulate=True, dyn=plan.total, dyn_dim=1)
        else:
            ops.gemm(th, mod.affine1.weight, dx, M=cap, N=F, K=A, lda=A, ldb=F, ldc=F, trans_b=True, accumulate=True, dyn=plan.total, dyn_dim=1)
        return dx, None, None, None


CAPTURE_RELU = [None]      # diagnostics / parity tests: set CAPTURE_RELU[0] = [] and every LinearFn with a ReLU appends its relu output r


class LinearFn(torch.autograd.Function):
    """y = dropout(act(x W^T + b)) for 2-D contiguous x;  act in {none, relu};  dropout mask keyed by the output element."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, p, seed):
        x = x.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        y = torch.empty((M, N), device=x.device, dtype=torch.float32)
        r = torch.empty((M, N), device=x.device, dtype=torch.float32) if (act == ops.ACT_RELU) else None
        ops.gemm(x, weight, y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, act=act, aux_out=r, ldaux=N, drop=(3, p, seed, N))
        if CAPTURE_RELU[0] is not None and r is not None:
            CAPTURE_RELU[0].append(r)
        ctx.x, ctx.weight, ctx.bias, ctx.r, ctx.act, ctx.p, ctx.seed = x, weight, bias, r, act, p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        if ctx.act == ops.ACT_RELU or ctx.p > 0:
            dz = torch.empty_like(dy)
            r = ctx.r if ctx.r is not None else torch.ones_like(dy)
            ops.relu_drop_bwd(dy, r, dz, None, ctx.p, ctx.seed)
        else:
            dz = dy
        gw, gb, x = grad_of(ctx.weight), (grad_of(ctx.bias) if ctx.bias is not None else None), ctx.x
        ops.leaf_deferred(dz.device, dz.shape[0], lambda: ops.linear_bwd_weight(dz, x, gw, db=gb), dz, x)
        dx = ops.linear_bwd_data(dz, ctx.weight)
        return dx, None, None, None, None, None


class QKVFn(torch.autograd.Function):
    """[Q | K | V] = x W_{Q,K,V}^T + b  into one [M, 3*h*d] buffer (layers.py:134-136).  When the three weights sit back to
    back in the flat parameter buffer (trainer.FlatParams honours MultiHeadAttention.adjacent_parameter_groups) they ARE one
    [3*h*d, d_model] matrix: one GEMM forward, one K = 3*h*d GEMM for dX (no read-modify-write of dX), one for dW, one bias
    reduction.  Otherwise (module used on its own) the same three-GEMM sequence per projection."""

    @staticmethod
    def _stacked(mha, grads=False):
        from .layers import stacked_view
        ws, bs = (mha.W_Q.weight, mha.W_K.weight, mha.W_V.weight), (mha.W_Q.bias, mha.W_K.bias, mha.W_V.bias)
        if grads:
            ws, bs = [grad_of(w) for w in ws], [grad_of(b) for b in bs]
        w, b = stacked_view(ws), stacked_view(bs)
        return (w, b) if w is not None and b is not None else (None, None)

    @staticmethod
    def forward(ctx, x, mha, dyn=None):
        """dyn: device int32 with the number of LIVE rows of x (packed token rows: the rest of the buffer is undefined)."""
        x = x.contiguous()
        M, K = x.shape
        HD = mha.W_Q.weight.shape[0]
        qkv = torch.empty((M, 3 * HD), device=x.device, dtype=torch.float32)
        w, b = QKVFn._stacked(mha)
        dk = dict(dyn=dyn, dyn_dim=1) if dyn is not None else {}
        if w is not None:
            ops.gemm(x, w, qkv, M=M, N=3 * HD, K=K, lda=K, ldb=K, ldc=3 * HD, bias=b, **dk)
        else:
            for s, lin in enumerate((mha.W_Q, mha.W_K, mha.W_V)):
                ops.gemm(x, lin.weight, qkv[:, s * HD:], M=M, N=HD, K=K, lda=K, ldb=K, ldc=3 * HD, bias=lin.bias, **dk)
        ctx.x, ctx.mha, ctx.HD, ctx.dyn = x, mha, HD, dyn
        return qkv

    @staticmethod
    def backward(ctx, dqkv):
        dqkv = dqkv.contiguous()
        x, mha, HD, dyn = ctx.x, ctx.mha, ctx.HD, ctx.dyn
        M, K = x.shape
        dx = torch.empty_like(x)
        w, _ = QKVFn._stacked(mha)
        gw, gb = QKVFn._stacked(mha, grads=True) if w is not None else (None, None)
        d1 = dict(dyn=dyn, dyn_dim=1) if dyn is not None else {}        # live rows bound M of the data gradient ...
        d2 = dict(dyn=dyn, dyn_dim=2) if dyn is not None else {}        # ... and the reduction of the weight gradient
        if gw is not None:
            # dW (+ the bias gradient, fused into the same launch) is a leaf: own stream, joined at the end of the pass
            ops.leaf_deferred(x.device, M, lambda: ops.gemm(dqkv, x, gw, M=3 * HD, N=K, K=M, lda=3 * HD, ldb=K, ldc=K, trans_a=True,
                                                         trans_b=True, split_k=ops.split_for(3 * HD, K, M, *ops.tn_tile(3 * HD, K, M)[1:]), atomic=True,
                                                         colsum_out=gb, tile=ops.tn_tile(3 * HD, K, M)[0], **d2), dqkv, x)
            if ops.USE_WT and M >= 1024 and w.is_contiguous():
                ops.gemm(dqkv, ops.wt(w), dx, M=M, N=K, K=3 * HD, lda=3 * HD, ldb=3 * HD, ldc=K, **d1)          # NT on [W_Q; W_K; W_V]^T
            else:
                ops.gemm(dqkv, w, dx, M=M, N=K, K=3 * HD, lda=3 * HD, ldb=K, ldc=K, trans_b=True, **d1)
            return dx, None, None
        for s, lin in enumerate((mha.W_Q, mha.W_K, mha.W_V)):
            d = dqkv[:, s * HD:]
            ops.gemm(d, lin.weight, dx, M=M, N=K, K=HD, lda=3 * HD, ldb=K, ldc=K, trans_b=True, accumulate=(s > 0), **d1)
            ops.gemm(d, x, grad_of(lin.weight), M=HD, N=K, K=M, lda=3 * HD, ldb=K, ldc=K, trans_a=True, trans_b=True,
                     split_k=ops.split_for(HD, K, M), atomic=True, **d2)
            ops.bias_grad(d, grad_of(lin.bias), rows=M, **({'dyn': dyn} if dyn is not None else {}))
        return dx, None, None


class MhsaCoreFn(torch.autograd.Function):
    """softmax(mask(Q K^T / sqrt(d_k))) V per head on the MFMA kernel (layers.py:137-147); p > 0 also applies the dropout
    that follows it (newsEncoders.py:196) in the kernel's output stage, same mask as DropoutFn(p, seed)."""

    @staticmethod
    def forward(ctx, qkv, mask, n, Lq, heads, dh, p=0.0, seed=0):
        qkv = qkv.contiguous()
        out = torch.empty((n * Lq, heads * dh), device=qkv.device, dtype=torch.float32)
        # the probabilities are not saved: backward recomputes them from Q, K (4 KB per head less HBM traffic each way)
        ops.mhsa_fwd(qkv, mask, n, Lq, heads, dh, out, None, p, seed)
        ctx.qkv, ctx.prob, ctx.mask, ctx.dims, ctx.drop = qkv, None, mask, (n, Lq, heads, dh), (p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        n, Lq, heads, dh = ctx.dims
        dqkv = torch.empty_like(ctx.qkv)
        ops.mhsa_bwd(ctx.qkv, ctx.mask, ctx.prob, dout.contiguous(), n, Lq, heads, dh, dqkv, *ctx.drop)
        return dqkv, None, None, None, None, None, None, None


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.dropout(x.contiguous(), p, seed)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(dy.contiguous(), ctx.p, ctx.seed), None, None


class Conv1dReluFn(torch.autograd.Function):
    """relu(Conv1d(E -> C, kernel k, 'same' padding)) over [n, L, E] (layers.py:33-35) as k shifted GEMMs:
    y[(i,t), :] = sum_dt x[(i, t+dt), :] . W[:, :, dt]^T  (rows outside [0, L) read as zero through the row-gather index)."""

    @staticmethod
    def forward(ctx, x, conv, n, Lx):
        x = x.contiguous()
        E = x.shape[1]
        Cn, _, k = conv.weight.shape
        pad = (k - 1) // 2
        dev = x.device
        wt = torch.empty((k, Cn, E), device=dev, dtype=torch.float32)              # [k][C][E]  (Conv1d stores [C][E][k])
        ops.transpose2d(conv.weight, wt, Cn * E, k)
        t = torch.arange(Lx, device=dev, dtype=torch.int32)
        base = (torch.arange(n, device=dev, dtype=torch.int32) * Lx)[:, None]
        idxs = []
        for j in range(k):
            tt = t + (j - pad)
            idxs.append(torch.where((tt >= 0) & (tt < Lx), base + tt[None, :], torch.full_like(base + tt[None, :], -1)).reshape(-1).contiguous())
        M = n * Lx
        y = torch.empty((M, Cn), device=dev, dtype=torch.float32)
        for j in range(k):
            last = j == k - 1
            ops.gemm(x, wt[j], y, M=M, N=Cn, K=E, lda=E, ldb=E, ldc=Cn, a_idx=idxs[j], accumulate=(2 if j > 0 else 0),
                     bias=conv.bias if last else None, act=ops.ACT_RELU if last else 0)
        ctx.x, ctx.conv, ctx.wt, ctx.idxs, ctx.y, ctx.dims = x, conv, wt, idxs, y, (n, Lx, E, Cn, k)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, Lx, E, Cn, k = ctx.dims
        M = n * Lx
        dz = ops.relu_bwd(dy.contiguous(), ctx.y)
        dx = torch.zeros_like(ctx.x)
        x, idxs, gw, gb = ctx.x, ctx.idxs, grad_of(ctx.conv.weight), grad_of(ctx.conv.bias)

        def weight_grads():              # leaves (ops.leaf_deferred); the bias gradient rides on the first launch's A tiles
            dwt = torch.zeros_like(ctx.wt)
            for j in range(k):
                ops.gemm(dz, x, dwt[j], M=Cn, N=E, K=M, lda=Cn, ldb=E, ldc=E, trans_a=True, trans_b=True, b_idx=idxs[j],
                         split_k=ops.split_for(Cn, E, M), atomic=True, colsum_out=gb if j == 0 else None)
            ops.transpose2d(dwt, gw, k, Cn * E, accumulate=True)
        ops.leaf_deferred(dz.device, M, weight_grads, dz, x)
        for j in range(k):
            # dx[(i, t+dt)] += dz[(i,t)] . W_dt : scattered to the shifted row (a bijection on valid rows -> plain accumulate)
            ops.gemm(dz, ctx.wt[j], dx, M=M, N=E, K=Cn, lda=Cn, ldb=E, ldc=E, trans_b=True, c_idx=ctx.idxs[j], accumulate=True)
        return dx, None, None, None


class FuseFn(torch.autograd.Function):
    """feature_fusion (newsEncoders.py:50-54): [rep | dropout(category row) | dropout(subCategory row)]."""

    @staticmethod
    def forward(ctx, rep, enc, category, subCategory, p, seed):
        n, F = rep.shape
        cd, sd = enc.category_embedding.weight.shape[1], enc.subCategory_embedding.weight.shape[1]
        D = F + cd + sd
        out = torch.empty((n, D), device=rep.device, dtype=torch.float32)
        ops.add2d(out, D, rep.contiguous(), F, n, F)
        cat = category.reshape(n).to(torch.int32).contiguous()
        sub = subCategory.reshape(n).to(torch.int32).contiguous()
        # (both tables in one launch, the kernel of the CNE step; same per-element masks as two nnr_small_embed_fwd calls)
        ops.fusion_rows_fwd(enc.category_embedding.weight, enc.subCategory_embedding.weight, cat, sub, None, None, out[:, F:], D, p, seed + 3, seed + 4)
        ctx.enc, ctx.cat, ctx.sub, ctx.dims, ctx.p, ctx.seed = enc, cat, sub, (n, F, cd, sd, D), p, seed
        return out

    @staticmethod
    def backward(ctx, dout):
        n, F, cd, sd, D = ctx.dims
        dout = dout.contiguous()
        ops.fusion_rows_bwd(ctx.cat, ctx.sub, None, None, cd, sd, dout[:, F:], D, grad_of(ctx.enc.category_embedding.weight),
                            grad_of(ctx.enc.subCategory_embedding.weight), ctx.p, ctx.seed + 3, ctx.seed + 4)
        drep = torch.empty((n, F), device=dout.device, dtype=torch.float32)
        ops.add2d(drep, F, dout, D, n, F)
        return drep, None, None, None, None, None


class ExpandFn(torch.autograd.Function):
    """[B, D] -> [B, N, D] (repeat / expand over the candidates, userEncoders.py:172,190); backward sums over N."""

    @staticmethod
    def forward(ctx, x, N):
        ctx.N = N
        return ops.expand_rows(x.contiguous(), N)

    @staticmethod
    def backward(ctx, dout):
        return ops.expand_rows_bwd(dout.contiguous()), None
