"""Evaluation path on the device: util.compute_scores (util.py:10-68) + evaluate.scoring
(evaluate.py:32-89).  Dev / test samples are (impression, candidate) pairs; the model scores each with news_num = 1
(util.py:43-50), candidates are ranked inside their impression and AUC / MRR / nDCG@5 / nDCG@10 are averaged over
impressions.  Batches come from a DeviceCorpus (id-only), scores never leave HBM until the four means are read back."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .corpus import DeviceCorpus
from .ops import _p, _s


def dev_corpus(arrays, device, category_num, graph='build', norm='symmetric'):
    """DeviceCorpus over dev / test behaviours (MIND_corpus.py:355-371): `arrays` as for DeviceCorpus plus beh_candidate [n]."""
    dc = DeviceCorpus(arrays, device, category_num, graph=graph, norm=norm)
    dc.set_samples(np.asarray(arrays['beh_candidate'], dtype=np.int32).reshape(-1, 1))
    return dc


@torch.no_grad()
def compute_scores(model, corpus, batch_size):
    """One score per dev sample, in dataset order (util.py:13-49): float32 [n] on the device."""
    was_training = model.training
    model.eval()
    scores = torch.zeros(corpus.num, device=corpus.device, dtype=torch.float32)
    for start in range(0, corpus.num, batch_size):
        idx = torch.arange(start, min(start + batch_size, corpus.num), device=corpus.device, dtype=torch.int32)
        batch = corpus.train_batch(idx)                       # candidate fields are [B, 1, ...] = the unsqueeze of util.py:43-48
        scores[start:start + idx.numel()] = model(*batch).squeeze(dim=1)
    model.train(was_training)
    return scores


def rank_metrics(scores, labels, sizes):
    """scores float32 [n], labels uint8/bool [n], sizes: candidates per impression (contiguous, file order).
    Returns (ranks int32 [n], per_impression float64 [n_imp, 4], means float64 [4]) -- all device tensors."""
    dev = scores.device
    sizes = torch.as_tensor(np.asarray(sizes), dtype=torch.int64)
    offsets = torch.zeros(sizes.numel() + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(sizes, 0)
    assert int(offsets[-1]) == scores.numel() == labels.numel()
    offsets = offsets.to(dev)
    lab = labels.to(device=dev, dtype=torch.uint8).contiguous()
    ranks = torch.empty(scores.numel(), device=dev, dtype=torch.int32)
    per = torch.empty((sizes.numel(), 4), device=dev, dtype=torch.float64)
    L.check(L.lib().nnr_rank_metrics(_p(scores.contiguous()), _p(lab), _p(offsets), sizes.numel(), _p(ranks), _p(per), _s()), 'nnr_rank_metrics')
    return ranks, per, per.mean(dim=0)


def evaluate(model, corpus, labels, sizes, batch_size):
    """compute_scores + scoring: -> (auc, mrr, ndcg5, ndcg10) as Python floats, plus the ranks (for the result file)."""
    scores = compute_scores(model, corpus, batch_size)
    ranks, _, means = rank_metrics(scores, torch.as_tensor(np.asarray(labels)), sizes)
    return tuple(float(v) for v in means.cpu()), ranks


def write_result_file(path, ranks, sizes):
    """The submission format of util.py:53-60: line i = '<i> [rank of candidate 0, rank of candidate 1, ...]'."""
    r = ranks.cpu().numpy().tolist()
    o = 0
    with open(path, 'w', encoding='utf-8') as f:
        for i, n in enumerate(sizes):
            f.write(('' if i == 0 else '\n') + str(i + 1) + ' ' + str(r[o:o + n]).replace(' ', ''))
            o += n
