"""Inner training step of the reference's `Trainer.train` / `distributed_train` (trainer.py:81-120, 261-300):
forward, loss, zero_grad, backward, [gradient all-reduce], clip_grad_norm_(gradient_clip_norm), Adam.step -- with the
optimizer state in flat fp32 buffers and clip+Adam fused into one HBM-bound kernel.  The epoch loop, dev evaluation and
checkpoint bookkeeping of the reference are control plane and out of scope."""
import torch
import torch.nn as nn

from . import dp, ops
from .layers import PARAM_EPOCH
from .model import negative_log_softmax


class FlatParams:
    """Re-homes every parameter of `model` (and its gradient) as a view into one flat fp32 buffer.
    Offsets are padded to 4 floats so every view is 16-byte aligned.  Device-agnostic (used by the gloo CPU tests too)."""

    def __init__(self, model: nn.Module):
        params = [p for p in model.parameters() if p.requires_grad]          # nn.Module dedups the shared news encoder
        # groups a module wants back to back (MultiHeadAttention: W_Q | W_K | W_V as one matrix) go first, in group order;
        # group members are multiples of 4 floats or the padding below would separate them (then the fused view is simply
        # not formed and the per-tensor path runs)
        grouped, seen = [], set()
        for m in model.modules():
            for g in getattr(m, 'adjacent_parameter_groups', lambda: [])():
                if all(q.requires_grad and id(q) not in seen for q in g):
                    grouped += g
                    seen.update(id(q) for q in g)
        params = grouped + [p for p in params if id(p) not in seen]
        self.params = params
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        dev = params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        for p, o in zip(params, offs):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
        self.numel = total
        self.offsets = offs

    def zero_grad(self):
        self.grad.zero_()


class _Own:
    """The parameters a user encoder owns itself: its direct parameters and its children except the shared news encoder."""

    def __init__(self, ue, children):
        self.ue, self.children = ue, children

    def parameters(self):
        for p in self.ue.parameters(recurse=False):
            yield p
        for m in self.children:
            yield from m.parameters()


class Trainer:
    def __init__(self, model, config):
        self.model = model
        self.config = config
        self.flat = FlatParams(model)
        self.m = torch.zeros_like(self.flat.flat)
        self.v = torch.zeros_like(self.flat.flat)
        self.sumsq = torch.zeros(1, device=self.flat.flat.device, dtype=torch.float32)
        self.step_count = 0
        self.gradient_clip_norm = float(config.gradient_clip_norm)
        self.lr, self.weight_decay = float(config.lr), float(config.weight_decay)
        dp.broadcast_parameters(self.flat.flat)
        # bucketed exchange: the user encoder's own gradients are final first (its backward runs before the news encoder's)
        ue = getattr(model, 'user_encoder', None)
        own = [m for name, m in ue.named_children() if name != 'news_encoder'] if ue is not None else []
        ne = getattr(model, 'news_encoder', None)
        table = ne.word_embedding.weight if (ne is not None and hasattr(ne, 'word_embedding')) else None
        self.exchange = dp.GradientExchange(self.flat, early_modules=[_Own(ue, own)] if ue is not None else [], table_param=table)
        if ue is not None:
            ue.__dict__['_grads_ready_hook'] = self.exchange.early_ready
        if ne is not None:
            ne.__dict__['_table_scatter_hook'] = self.exchange.table_scatter_done      # (CNE calls it after each embedding-row scatter GEMM)

    def train_step(self, batch):
        """One optimizer step on `batch` (21 device tensors, Model.forward order).  Returns (logits, loss) as device
        tensors -- no host synchronisation (the reference's float(loss) sync at trainer.py:115 is the caller's choice)."""
        model = self.model
        self.flat.zero_grad()
        logits = model(*batch)
        loss = negative_log_softmax(logits)
        loss.backward()
        ops.join_extra_streams()
        scale = self.exchange.finish()
        self.optimizer_step(scale)
        return logits.detach(), loss.detach()

    def optimizer_step(self, grad_scale=1.0):
        self.step_count += 1
        self.sumsq.zero_()
        ops.sumsq(self.flat.grad, self.sumsq)
        ops.clip_adam(self.flat.flat, self.flat.grad, self.m, self.v, self.sumsq, grad_scale, self.gradient_clip_norm, self.lr, 0.9, 0.999,
                      1e-8, self.weight_decay, self.step_count)
        PARAM_EPOCH[0] += 1           # parameters changed behind torch's back: invalidate cached weight layouts

    def grad_total_norm(self, grad_scale=1.0):
        s = torch.zeros(1, device=self.flat.grad.device, dtype=torch.float32)
        ops.sumsq(self.flat.grad, s)
        return float(s.sqrt()) * grad_scale
