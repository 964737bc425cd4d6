"""The CNE+SUE training step (trainer.py:105-120 over model.py:120-133) written out as ONE sequence of C-ABI calls, without an
autograd graph: candidate + history encoder call (one packed token stream) -> SUE -> click predictor + loss + d logits in one
kernel -> click-predictor backward -> SUE backward -> news-encoder backward -> [gradient exchange] -> clip + Adam.

Same kernels, same order, same HIP streams and the same dropout seeds as `loss.backward()` through the autograd Functions of
news_encoders / user_encoders / model (tests/test_hip_tape_gpu.py compares the two); what is gone is the framework in between:
autograd's bookkeeping, its gradient-accumulation / fill / cat kernels, and every host-side tensor op that is not a call into
libnnr_hip.so.  That makes the step RECORDABLE: nnr_amd.tape captures the calls of one such step and replays them natively."""
import contextlib
import os

import torch

from . import ops
from .news_encoders import cne_forward_many, cne_backward_many, _CNE_UNION

_BX3_MHSA = os.environ.get('NNR_BX3_MHSA', '0') == '1'              # A/B: every class of ops._BX3_CLASSES in the MHSA step too
_CNE_STEP_ROWS = os.environ.get('NNR_CNE_STEP_ROWS', '0') == '1'      # A/B: post the history call's token rows (Model.forward's rule) instead of 0
_BX3_MIN_SEQS = int(os.environ.get('NNR_BX3_MIN_SEQS', '1408'))         # CNE + SUE: bf16x3 from this many news-encoder sequences per step on (batch 32: 1 760)
_BX3_MHSA_CLASSES = set(c for c in os.environ.get('NNR_BX3_MHSA_CLASSES', 'dx').split(',') if c)
_MHSA_NATIVE = os.environ.get('NNR_MHSA_NATIVE', '1') != '0'      # A/B: MHSA+MHSA through autograd (round 3) instead of the native step


def kind(model):
    """Which native step covers `model`: 'cne_sue' (the headline pair, BASELINE.json configs[2..4]; device-side tie order), 'mhsa'
    (configs[1]: MHSA news + MHSA user encoder), or None (autograd path)."""
    from . import news_encoders as NE, user_encoders as UE
    if model.click_predictor != 'dot_product' or not model.training:
        return None
    if type(model.news_encoder) is NE.CNE and type(model.user_encoder) is UE.SUE and model.news_encoder.tie_order == 'stable' and NE._CNE_UNION:
        return 'cne_sue'
    if type(model.news_encoder) is NE.MHSA and type(model.user_encoder) is UE.MHSA and _MHSA_NATIVE:
        return 'mhsa'
    return None


def step_sequences(batch):
    """Sequences the news encoder runs in one step on `batch` (candidates + history: 3 520 at per-GPU batch 64)."""
    return int(batch[15].shape[0]) * (int(batch[15].shape[1]) + int(batch[3].shape[1]))


def bx3_classes(model, seqs=None):
    """The shape classes (ops.bx3_class) whose weight-operand NT GEMMs run on the bf16x3 kernel in a step of `model`; empty = the pure fp32-MFMA
    path.  CNE + SUE: every class (ops._BX3_CLASSES) -- from `_BX3_MIN_SEQS` news-encoder sequences per step on (`seqs`: step_sequences(batch);
    None = unknown, no size rule): at per-GPU batch 8 / 16 (440 / 880 sequences, the 8- and 4-GPU shards of a global batch of 64) the products are
    too small to repay the step's ~33 image splits and the 63 KB tiles -- fp32 kernels 3.20 vs 3.37 ms and 4.31 vs 4.45 ms; batch 32 (1 760): bx3
    5.70-5.80 vs 5.93-6.02, batch 64: 9.45 vs 10.1 (profiles/r06_ab.txt calls 34-36).  MHSA news encoder (configs[1]): only the K >= 1024
    data-gradient products ('dx': dQKV of the user encoder, the word-embedding gradient's 3 h d -> E product) -- per-class same-box A/Bs, call 30:
    'dx' -0.03..-0.07 ms on three pairs, 'proj' neutral, 'gate' and all classes together unstable (the 63 KB tiles beside the attention core's
    workgroups)."""
    from . import news_encoders as NE
    if not ops.BX3[0]:
        return set()
    if type(getattr(model, 'news_encoder', None)) is NE.MHSA:
        return set(ops._BX3_CLASSES) if _BX3_MHSA else set(_BX3_MHSA_CLASSES)
    if seqs is not None and seqs < _BX3_MIN_SEQS:
        return set()
    return set(ops._BX3_CLASSES)


@contextlib.contextmanager
def matrix_path(model, batch=None):
    """The matrix path of one step of `model` on `batch` (bx3_classes).  Used by the native step AND by the trainer's autograd path, so that
    both run the same kernels."""
    want = bx3_classes(model, None if batch is None else step_sequences(batch))
    on, classes = ops.BX3[0], ops._BX3_CLASSES
    if on:
        if want:
            ops._BX3_CLASSES = want
        else:
            ops.BX3[0] = False
    try:
        yield
    finally:
        ops.BX3[0], ops._BX3_CLASSES = on, classes


def supported(model):
    return kind(model) is not None


def news_calls_per_step(model):
    """Encoder calls (= per-call dropout seeds drawn) of one step: the CNE step plans candidates + history as ONE call, the MHSA step
    makes the reference's two calls (model.py:123-125)."""
    return 2 if kind(model) == 'mhsa' else 1


_ID_INPUTS = (1, 2, 3, 6, 13, 14, 15, 18)       # category / subCategory / title / content ids of the history and the candidate call


def recordable(batch):
    """May a launch tape be recorded from a step on `batch`?  Every tensor must be used where it lies: a non-contiguous tensor or an
    int64 id tensor would be converted by a torch op outside the library (news_encoders._i32 / .contiguous()) into a temporary the
    tape knows nothing about.  Such batches (not what the reference's DataLoader or DeviceCorpus produce: SURVEY.md appendix C) run
    the native step call by call."""
    return (all(t.is_cuda and t.is_contiguous() for t in batch) and all(batch[i].dtype == torch.int32 for i in _ID_INPUTS)
            and batch[11].dim() == 2 and batch[11].element_size() == 1 and batch[12].dtype == torch.int64)


class _Ctx:
    """Stands in for autograd's ctx: the encoders' building blocks (nnr_amd.functional, layers._AttentionFn) are autograd Functions whose
    static forward / backward only set and read attributes on it."""

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def _fwd(fn, *args):
    ctx = _Ctx()
    return fn.forward(ctx, *args), (fn, ctx)


def _bwd(node, *grads):
    fn, ctx = node
    return fn.backward(ctx, *grads)


def _mhsa_news_forward(ne, title_text, title_mask, category, subCategory):
    """newsEncoders.py:187-200 (nnr_amd.news_encoders.MHSA.forward) as a plain call sequence; returns ([B, N, D], nodes)."""
    from . import functional as Fn
    from .layers import _AttentionFn
    B, N = title_text.shape[:2]
    n, Lx = B * N, ne.max_sentence_length
    p = ne.dropout_rate if ne.training else 0.0
    seed = ne._next_seed()
    mask = title_mask.view(n, Lx)
    from .news_encoders import mhsa_packed
    if mhsa_packed(ne, title_text):
        pack = Fn.MhsaPack(mask, title_text.reshape(n, Lx))
        w, n1 = _fwd(Fn.PackedEmbedDropFn, ne.word_embedding.weight, pack, p, seed + 1)
        qkv, n2 = _fwd(Fn.QKVFn, w, ne.multiheadAttention, pack.plan.total)
        c, n3 = _fwd(Fn.PackedMhsaCoreFn, qkv, mask, pack, ne.head_num, ne.head_dim, p, seed + 2)
        rep, n4 = _fwd(Fn.PackedAttentionFn, c, ne.attention, mask, pack)
        out, n5 = _fwd(Fn.FuseFn, rep, ne, category, subCategory, p, seed)
        return out.view(B, N, ne.news_embedding_dim), (n1, n2, n3, n4, n5, (n, Lx, True))
    w, n1 = _fwd(Fn.EmbedDropFn, ne.word_embedding.weight, title_text, p, seed + 1)
    qkv, n2 = _fwd(Fn.QKVFn, w, ne.multiheadAttention)
    c, n3 = _fwd(Fn.MhsaCoreFn, qkv, mask, n, Lx, ne.head_num, ne.head_dim, p, seed + 2)
    rep, n4 = _fwd(_AttentionFn, c.view(n, Lx, ne.feature_dim), ne.attention, mask)
    out, n5 = _fwd(Fn.FuseFn, rep, ne, category, subCategory, p, seed)
    return out.view(B, N, ne.news_embedding_dim), (n1, n2, n3, n4, n5, (n, Lx, False))


def _mhsa_news_backward(ne, nodes, dout):
    n1, n2, n3, n4, n5, (n, Lx, packed) = nodes
    drep = _bwd(n5, dout.reshape(n, -1))[0]
    dc = _bwd(n4, drep)[0]
    dqkv = _bwd(n3, dc if packed else dc.reshape(n * Lx, -1))[0]
    dw = _bwd(n2, dqkv)[0]
    _bwd(n1, dw)


def forward_backward_mhsa(trainer, batch):
    """BASELINE.json configs[1] (MHSA + MHSA) without an autograd graph: the same building blocks, seeds, HIP streams and order as
    Model.forward + loss.backward() (candidate call on the side stream beside the history call; weight gradients on the leaf stream),
    every device-side operation a C-ABI call -- so the step is recordable (nnr_amd.tape) like the CNE+SUE one."""
    from . import functional as Fn
    from .layers import _AttentionFn
    from .news_encoders import _side_stream
    model = trainer.model
    ne, ue = model.news_encoder, model.user_encoder
    (user_ID, user_category, user_subCategory, user_title_text, user_title_mask, user_title_entity, user_content_text, user_content_mask,
     user_content_entity, user_history_mask, user_history_graph, user_history_category_mask, user_history_category_indices, news_category,
     news_subCategory, news_title_text, news_title_mask, news_title_entity, news_content_text, news_content_mask, news_content_entity) = batch
    dev = news_title_text.device
    f32 = dict(device=dev, dtype=torch.float32)
    with torch.no_grad():
        ops.wt_prefetch(dev)
        ops.bx3_prefetch(dev)                             # the bf16 images of the transposes above, same leaf stream (idle at the head of this step)
        ops.STEP_ROWS[0] = user_title_text.shape[0] * user_title_text.shape[1] * user_title_text.shape[2]
        ops._DEFER['manual'] = True                      # ops.leaf_deferred: no autograd end-of-pass callback here; joined below
        try:
            side, main = _side_stream(dev), torch.cuda.current_stream(dev)
            two = ops.SIDE_CALL and ops.STEP_ROWS[0] >= ops.LEAF_MIN_ROWS and not ops.ONE_STREAM[0]
            if two:                                       # the candidate call on a side stream beside the history call (model.py:123-125)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    rep_c, nodes_c = _mhsa_news_forward(ne, news_title_text, news_title_mask, news_category, news_subCategory)
            else:
                rep_c, nodes_c = _mhsa_news_forward(ne, news_title_text, news_title_mask, news_category, news_subCategory)
            rep_h, nodes_h = _mhsa_news_forward(ne, user_title_text, user_title_mask, user_category, user_subCategory)
            if two:
                main.wait_stream(side)
            B, N, D = rep_c.shape
            Hn = rep_h.shape[1]
            # user encoder (userEncoders.py:164-173)
            qkv, u1 = _fwd(Fn.QKVFn, rep_h.reshape(B * Hn, D), ue.multiheadAttention)
            h, u2 = _fwd(Fn.MhsaCoreFn, qkv, user_history_mask.contiguous(), B, Hn, ue.head_num, ue.head_dim)
            h, u3 = _fwd(Fn.LinearFn, h, ue.affine.weight, ue.affine.bias, ops.ACT_RELU, 0.5 if ue.training else 0.0, ue._next_seed())
            user, u4 = _fwd(_AttentionFn, h.view(B, Hn, D), ue.attention, None)
            user_rep, u5 = _fwd(Fn.ExpandFn, user, N)
            # click predictor + loss + their backward in one launch (model.py:126-127, trainer.py:64-66)
            logits = torch.empty((B, N), **f32)
            loss = torch.empty((), **f32)
            duser = torch.empty((B, N, D), **f32)
            dcand = torch.empty((B, N, D), **f32)
            trainer.wait_grad_zeroed()
            ops.click_loss(user_rep, rep_c.contiguous(), B, N, D, logits, loss, None, duser, dcand, torch.empty(B, **f32))
            # backward: user encoder, then the two encoder calls (the candidates' on the side stream again)
            du = _bwd(u5, duser)[0]
            dh = _bwd(u4, du)[0]
            dh = _bwd(u3, dh.reshape(B * Hn, D))[0]
            dqkv = _bwd(u2, dh)[0]
            dhist = _bwd(u1, dqkv)[0]
            hook = ue.__dict__.get('_grads_ready_hook')
            if hook is not None:
                # The user encoder's weight gradients were issued on the leaf stream (ops.leaf_deferred under _DEFER['manual']) and are
                # joined only at the end of the step.  The early bucket's all-reduce is ordered behind THIS stream only, so under data
                # parallelism this stream first waits for the leaf streams (round-4 advisor, high: otherwise the collective could read
                # -- and the leaf GEMMs / slab reductions later add into -- a half-written bucket).  sue_backward does the same (need_now).
                if trainer.exchange.active():
                    ops.join_leaf_streams(dev)
                hook()
            if two:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    _mhsa_news_backward(ne, nodes_c, dcand)
                _mhsa_news_backward(ne, nodes_h, dhist.reshape(B, Hn, D))
                main.wait_stream(side)
            else:
                _mhsa_news_backward(ne, nodes_c, dcand)
                _mhsa_news_backward(ne, nodes_h, dhist.reshape(B, Hn, D))
        finally:
            ops._DEFER['manual'] = False
        ops.join_extra_streams()
    return logits, loss


def forward_backward(trainer, batch):
    """Forward + loss + backward of one batch (21 device tensors, Model.forward order) into the trainer's flat gradient buffer.
    Returns (logits [B, N], loss []) -- fresh tensors of this call."""
    model = trainer.model
    if kind(model) == 'mhsa':
        # The MHSA + MHSA step keeps most of its GEMMs on the fp32-MFMA kernels: its products are small (~100 GFLOP per step, 40 k live title rows)
        # and interleaved with the attention-core launches; with EVERY class on the bf16x3 tiles (63 KB of LDS, two workgroups per CU) the step measured
        # slower on three same-box pairs (2.18 vs 2.24 ms, profiles/r06_ab.txt).  Only the K >= 1024 data-gradient products take them (bx3_classes).
        with matrix_path(model, batch):
            return forward_backward_mhsa(trainer, batch)
    ne, ue = model.news_encoder, model.user_encoder
    (user_ID, user_category, user_subCategory, user_title_text, user_title_mask, user_title_entity, user_content_text, user_content_mask,
     user_content_entity, user_history_mask, user_history_graph, user_history_category_mask, user_history_category_indices, news_category,
     news_subCategory, news_title_text, news_title_mask, news_title_entity, news_content_text, news_content_mask, news_content_entity) = batch
    dev = news_title_text.device
    f32 = dict(device=dev, dtype=torch.float32)
    ops._DEFER['step_joins'] = True                   # this function ends with join_extra_streams(): leaf work may stay un-joined until then
    try:
        with matrix_path(model, batch):
            return _forward_backward_cne_sue(trainer, model, ne, ue, batch, dev, f32)
    finally:
        ops._DEFER['step_joins'] = False


def _forward_backward_cne_sue(trainer, model, ne, ue, batch, dev, f32):
    (user_ID, user_category, user_subCategory, user_title_text, user_title_mask, user_title_entity, user_content_text, user_content_mask,
     user_content_entity, user_history_mask, user_history_graph, user_history_category_mask, user_history_category_indices, news_category,
     news_subCategory, news_title_text, news_title_mask, news_title_entity, news_content_text, news_content_mask, news_content_entity) = batch
    with torch.no_grad():
        # ops.leaf_deferred's "is the step big enough to defer small reductions" input.  This step never posted it and ran on whatever the process's last
        # OTHER step had left there (0 in a fresh process -- what every A/B of this step was tuned on --, 102 400 behind an MHSA step): posted now, so
        # that the step does not depend on its predecessors.  (Measured neutral either way, call 49; it is NOT why bench.py's batch-8 leg runs 7 % behind
        # its stand-alone command -- see ops.new_stream.)
        ops.STEP_ROWS[0] = user_title_text.numel() if _CNE_STEP_ROWS else 0
        ops.wt_prefetch(dev)                          # W^T copies the backward pass multiplies by, on the leaf stream
        cand = (news_title_text, news_title_mask, news_content_text, news_content_mask, news_category, news_subCategory)
        hist = (user_title_text, user_title_mask, user_content_text, user_content_mask, user_category, user_subCategory)
        ((rep_c, rep_h), sv), = cne_forward_many(ne, [tuple(zip(cand, hist))])
        B, N, D = rep_c.shape
        n0 = B * N
        n = n0 + rep_h.shape[0] * rep_h.shape[1]
        # user encoder (userEncoders.py:73-98)
        from .user_encoders import sue_forward, sue_backward
        assert user_history_graph.is_contiguous() and user_history_category_indices.is_contiguous()
        user, ssv = sue_forward(ue, rep_h, rep_c, user_history_graph, user_history_category_mask, user_history_category_indices)
        # click predictor + loss (model.py:126-127, trainer.py:64-66) and their backward in ONE launch.  The gradient of the union
        # stream's representations lives in one [n, D] buffer: rows [0, n0) the candidates (click predictor + user encoder), rows
        # [n0, n) the history news (user encoder)
        logits = torch.empty((B, N), **f32)
        loss = torch.empty((), **f32)
        drep = torch.empty((n, D), **f32)
        duser = torch.empty((B, N, D), **f32)
        trainer.wait_grad_zeroed()                     # the flat gradient buffer was cleared on the leaf stream while the forward pass ran
        ops.click_loss(user, rep_c, B, N, D, logits, loss, None, duser, drep[:n0], torch.empty(B, **f32))
        sue_backward(ue, ssv, duser, dhist_out=drep[n0:].view(B, -1, D), dcand_accum=drep[:n0])
        cne_backward_many(ne, [(sv, drep)])
        ops.join_extra_streams()
    return logits, loss
