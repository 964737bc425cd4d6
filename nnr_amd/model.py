"""`Model(news_encoder, user_encoder, click_predictor)` -- the reference's plugin hub (model.py:10-133) for the
in-scope encoders, dispatching on the same `--news_encoder / --user_encoder` strings; forward takes the same 21
positional tensors (trainer.py:105-106) and returns logits [batch, 1 + negative_sample_num]."""
import torch
import torch.nn as nn

from . import ops
from . import news_encoders as newsEncoders
from . import user_encoders as userEncoders


class _DotProductFn(torch.autograd.Function):
    """logits = (user_representation * news_representation).sum(dim=2)   (model.py:126-127)"""

    @staticmethod
    def forward(ctx, user, cand):
        B, N, D = cand.shape
        user, cand = user.contiguous(), cand.contiguous()
        logits = torch.empty((B, N), device=cand.device, dtype=torch.float32)
        ops.logits_fwd(user, cand, B, N, D, logits)
        ctx.save_for_backward(user, cand)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        user, cand = ctx.saved_tensors
        B, N, D = cand.shape
        duser, dcand = torch.empty_like(user), torch.empty_like(cand)
        ops.logits_bwd(dlogits.contiguous(), user, cand, B, N, D, duser, dcand)
        return duser, dcand


class _NegLogSoftmaxFn(torch.autograd.Function):
    """loss = (-log_softmax(logits, dim=1)[:, 0]).mean()   (trainer.py:64-66); the gradient is produced in the same pass."""

    @staticmethod
    def forward(ctx, logits):
        B, N = logits.shape
        logits = logits.contiguous()
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        dlogits = torch.empty_like(logits)
        ops.nls_loss(logits, B, N, loss, dlogits)
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dlogits,) = ctx.saved_tensors
        return dlogits * dloss


class _JoinSideFn(torch.autograd.Function):
    """Fork/join of the candidate encoder call that ran on a side HIP stream (autograd-based encoders, GPU-bound steps).
    forward: the current (main) stream waits for the side stream.  backward: autograd runs the candidate call's backward nodes
    on the side stream again (and orders them behind the producer of this gradient); the end-of-pass callback joins the side
    stream back, because those nodes write parameter gradients out of autograd's sight."""

    @staticmethod
    def forward(ctx, x, side):
        ctx.side, ctx.main = side, torch.cuda.current_stream(x.device)
        ctx.main.wait_stream(side)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        side = ctx.side
        g.record_stream(side)                        # consumed by side-stream kernels after this node's buffer is released
        torch.autograd.Variable._execution_engine.queue_callback(lambda: ops.join_extra_streams(g.device))
        return g, None


def negative_log_softmax(logits):
    return _NegLogSoftmaxFn.apply(logits)


class Model(nn.Module):
    def __init__(self, config, word_table=None):
        super().__init__()
        if config.news_encoder == 'CNE':
            self.news_encoder = newsEncoders.CNE(config, word_table)
        elif config.news_encoder == 'CNN':
            self.news_encoder = newsEncoders.CNN(config, word_table)
        elif config.news_encoder == 'MHSA':
            self.news_encoder = newsEncoders.MHSA(config, word_table)
        else:
            raise Exception(config.news_encoder + ' is not on the MI355X hot path (in scope: CNE, CNN, MHSA; SURVEY.md section 8a)')
        if config.user_encoder == 'SUE':
            self.user_encoder = userEncoders.SUE(self.news_encoder, config)
        elif config.user_encoder == 'MHSA':
            self.user_encoder = userEncoders.MHSA(self.news_encoder, config)
        elif config.user_encoder == 'ATT':
            self.user_encoder = userEncoders.ATT(self.news_encoder, config)
        else:
            raise Exception(config.user_encoder + ' is not on the MI355X hot path (in scope: SUE, MHSA, ATT; SURVEY.md section 8a)')
        self.model_name = config.news_encoder + '-' + config.user_encoder
        self.news_embedding_dim = self.news_encoder.news_embedding_dim
        self.dropout = nn.Dropout(p=config.dropout_rate)                    # (model.py:77: part of the attribute surface; only the
                                                                              #  out-of-scope mlp / FIM click predictors call it)
        self.use_user_embedding = False
        if config.click_predictor != 'dot_product':
            raise Exception('click_predictor=%s is out of scope (dot_product only, model.py:126-127)' % config.click_predictor)
        self.click_predictor = config.click_predictor

    def initialize(self):
        self.news_encoder.initialize()
        self.user_encoder.initialize()

    def forward(self, user_ID, user_category, user_subCategory, user_title_text, user_title_mask, user_title_entity, user_content_text,
                user_content_mask, user_content_entity, user_history_mask, user_history_graph, user_history_category_mask,
                user_history_category_indices, news_category, news_subCategory, news_title_text, news_title_mask, news_title_entity,
                news_content_text, news_content_mask, news_content_entity):
        user_embedding = None
        if self.training and torch.is_grad_enabled() and news_title_text.is_cuda:
            ops.wt_prefetch(news_title_text.device)      # W^T copies the backward pass will want, off the critical chain
        ne = self.news_encoder
        if (hasattr(ne, 'forward_pair') and not self.training and not torch.is_grad_enabled() and news_title_text.is_cuda
                and getattr(ne, 'pad_dedup', True) and ne.tie_order == 'stable' and hasattr(self.user_encoder, 'encode_user')):
            # inference: the history call without its redundant PAD slots (exact; news_encoders.cne_history_dedup)
            news_representation = ne(news_title_text, news_title_mask, news_title_entity, news_content_text, news_content_mask,
                                     news_content_entity, news_category, news_subCategory, user_embedding)
            history_embedding, rows = newsEncoders.cne_history_dedup(ne, user_title_text, user_title_mask, user_content_text, user_content_mask,
                                                                     user_category, user_subCategory)
            st = ne.__dict__.setdefault('_dedup_stats', [0, 0])
            st[0] += rows
            st[1] += user_title_text.shape[0] * user_title_text.shape[1]
            user_representation = self.user_encoder.encode_user(history_embedding, user_history_mask, user_history_graph,
                                                                user_history_category_mask, user_history_category_indices,
                                                                news_representation)
        elif hasattr(self.news_encoder, 'forward_pair'):
            # same arithmetic as the two encoder calls of model.py:123-125, issued in lock-step so that launch-latency-bound
            # stages (the Bi-LSTM recurrences) of the candidate call and of the history call share one launch
            news_representation, history_embedding = self.news_encoder.forward_pair(
                (news_title_text, news_title_mask, news_content_text, news_content_mask, news_category, news_subCategory),
                (user_title_text, user_title_mask, user_content_text, user_content_mask, user_category, user_subCategory))
            user_representation = self.user_encoder.encode_user(history_embedding, user_history_mask, user_history_graph,
                                                                user_history_category_mask, user_history_category_indices,
                                                                news_representation)
        else:
            # size class of this step for ops.leaf_deferred: the token rows of the history call
            ops.STEP_ROWS[0] = user_title_text.shape[0] * user_title_text.shape[1] * user_title_text.shape[2]
            cand = (news_title_text, news_title_mask, news_title_entity, news_content_text, news_content_mask, news_content_entity,
                    news_category, news_subCategory, user_embedding)
            if news_title_text.is_cuda and ops.SIDE_CALL and ops.STEP_ROWS[0] >= ops.LEAF_MIN_ROWS and hasattr(self.user_encoder, 'encode_user'):
                # GPU-bound step: the small candidate call runs on a side HIP stream next to the history call (same launches,
                # same seeds, same order of the dropout counters as the sequential form below)
                dev = news_title_text.device
                side, main = newsEncoders._side_stream(dev), torch.cuda.current_stream(dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    news_representation = self.news_encoder(*cand)
                history_embedding = self.news_encoder(user_title_text, user_title_mask, user_title_entity, user_content_text,
                                                      user_content_mask, user_content_entity, user_category, user_subCategory, user_embedding)
                news_representation = _JoinSideFn.apply(news_representation, side)
                user_representation = self.user_encoder.encode_user(history_embedding, user_history_mask, user_history_graph,
                                                                    user_history_category_mask, user_history_category_indices,
                                                                    news_representation)
            else:
                news_representation = self.news_encoder(*cand)
                user_representation = self.user_encoder(user_title_text, user_title_mask, user_title_entity, user_content_text,
                                                        user_content_mask, user_content_entity, user_category, user_subCategory,
                                                        user_history_mask, user_history_graph, user_history_category_mask,
                                                        user_history_category_indices, user_embedding, news_representation)
        return _DotProductFn.apply(user_representation, news_representation)
