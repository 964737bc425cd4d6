"""The CNE+SUE training step (trainer.py:105-120 over model.py:120-133) written out as ONE sequence of C-ABI calls, without an
autograd graph: candidate + history encoder call (one packed token stream) -> SUE -> click predictor + loss + d logits in one
kernel -> click-predictor backward -> SUE backward -> news-encoder backward -> [gradient exchange] -> clip + Adam.

Same kernels, same order, same HIP streams and the same dropout seeds as `loss.backward()` through the autograd Functions of
news_encoders / user_encoders / model (tests/test_hip_tape_gpu.py compares the two); what is gone is the framework in between:
autograd's bookkeeping, its gradient-accumulation / fill / cat kernels, and every host-side tensor op that is not a call into
libnnr_hip.so.  That makes the step RECORDABLE: nnr_amd.tape captures the calls of one such step and replays them natively."""
import torch

from . import ops
from .news_encoders import cne_forward_many, cne_backward_many, _CNE_UNION


def supported(model):
    """The native step covers the headline pair (BASELINE.json: CNE + SUE, dot-product click predictor), device-side tie order."""
    from . import news_encoders as NE, user_encoders as UE
    return (type(model.news_encoder) is NE.CNE and type(model.user_encoder) is UE.SUE and model.click_predictor == 'dot_product'
            and model.news_encoder.tie_order == 'stable' and NE._CNE_UNION and model.training)


_ID_INPUTS = (1, 2, 3, 6, 13, 14, 15, 18)       # category / subCategory / title / content ids of the history and the candidate call


def recordable(batch):
    """May a launch tape be recorded from a step on `batch`?  Every tensor must be used where it lies: a non-contiguous tensor or an
    int64 id tensor would be converted by a torch op outside the library (news_encoders._i32 / .contiguous()) into a temporary the
    tape knows nothing about.  Such batches (not what the reference's DataLoader or DeviceCorpus produce: SURVEY.md appendix C) run
    the native step call by call."""
    return (all(t.is_cuda and t.is_contiguous() for t in batch) and all(batch[i].dtype == torch.int32 for i in _ID_INPUTS)
            and batch[11].dim() == 2 and batch[11].element_size() == 1 and batch[12].dtype == torch.int64)


def forward_backward(trainer, batch):
    """Forward + loss + backward of one batch (21 device tensors, Model.forward order) into the trainer's flat gradient buffer.
    Returns (logits [B, N], loss []) -- fresh tensors of this call."""
    model = trainer.model
    ne, ue = model.news_encoder, model.user_encoder
    (user_ID, user_category, user_subCategory, user_title_text, user_title_mask, user_title_entity, user_content_text, user_content_mask,
     user_content_entity, user_history_mask, user_history_graph, user_history_category_mask, user_history_category_indices, news_category,
     news_subCategory, news_title_text, news_title_mask, news_title_entity, news_content_text, news_content_mask, news_content_entity) = batch
    dev = news_title_text.device
    f32 = dict(device=dev, dtype=torch.float32)
    with torch.no_grad():
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
