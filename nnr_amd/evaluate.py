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


LAST_STATS = {}          # filled by compute_scores: how many news rows the news encoder processed (cached vs per-sample form)


def news_reps_cacheable(model):
    """A news representation may be cached (encoded once per evaluation, gathered by id) only if it does not depend on the
    other news of its encoder call.  True for the MHSA / CNN encoders; NOT for CNE: the reference gates the title at
    title-rank r with the content memory at content-rank r of the SAME call (newsEncoders.py:112-115,128-129), so its
    representations depend on their batch-mates and caching would change the reference's scores."""
    return getattr(model.news_encoder, 'batch_independent', False) and hasattr(model.user_encoder, 'encode_user')


@torch.no_grad()
def encode_news_table(model, corpus, ids, chunk=4096):
    """Representations [len(ids), D] of the news `ids` (int64 device tensor of news indices), chunk by chunk, eval mode."""
    t = corpus.t
    ne = model.news_encoder
    out = torch.empty((ids.numel(), ne.news_embedding_dim), device=corpus.device, dtype=torch.float32)
    for s in range(0, ids.numel(), chunk):
        sel = ids[s:s + chunk]
        f = lambda k: t[k][sel].unsqueeze(1).contiguous()          # [n, 1, ...]: one news per "sample"
        rep = ne(f('news_title_text'), f('news_title_mask'), f('news_title_entity'), f('news_abstract_text'), f('news_abstract_mask'),
                 f('news_abstract_entity'), f('news_category'), f('news_subCategory'), None)
        out[s:s + sel.numel()] = rep.view(sel.numel(), -1)
    return out


@torch.no_grad()
def compute_scores_cached(model, corpus, batch_size):
    """compute_scores for encoders with batch-independent news representations (the reference's README
    lists the missing cache as a known cost, README.md:125): every DISTINCT news of the split is encoded once, a sample's history
    / candidate representations are gathered by news id, and only the user encoder + click predictor run per sample."""
    assert news_reps_cacheable(model)
    was_training = model.training
    model.eval()
    t, dev = corpus.t, corpus.device
    hist, cand = t['beh_history'].long(), corpus.samples[:, 0].long()
    used, inv = torch.unique(torch.cat([hist.reshape(-1), cand]), return_inverse=True)
    reps = encode_news_table(model, corpus, used)
    slot_h, slot_c = inv[:hist.numel()].view_as(hist), inv[hist.numel():]
    scores = torch.zeros(corpus.num, device=dev, dtype=torch.float32)
    ue = model.user_encoder
    from .model import _DotProductFn
    need_graph = type(ue).__name__ == 'SUE'
    for start in range(0, corpus.num, batch_size):
        idx = torch.arange(start, min(start + batch_size, corpus.num), device=dev)
        h = reps[slot_h[idx]]                                  # [B, H, D]
        c = reps[slot_c[idx]].unsqueeze(1)                     # [B, 1, D]
        hmask = t['beh_history_mask'][idx]
        graph = cmask = cidx = None
        if need_graph:
            from .corpus import history_graph
            graph, cmask, cidx = history_graph(t['news_category'][hist[idx]].contiguous(), hmask.contiguous(), corpus.category_num, corpus.norm_name)
        user = ue.encode_user(h, hmask, graph, cmask, cidx, c)
        scores[start:start + idx.numel()] = _DotProductFn.apply(user, c).squeeze(dim=1)
    model.train(was_training)
    LAST_STATS.update(mode='cached', encoder_rows=int(used.numel()), per_sample_rows=int(corpus.num * (hist.shape[1] + 1)))
    return scores


@torch.no_grad()
def compute_scores(model, corpus, batch_size, cache='auto'):
    """One score per dev sample, in dataset order (util.py:13-49): float32 [n] on the device.  cache: 'auto' (cache the news
    representations when the encoder allows it, see news_reps_cacheable), True, False (the reference's per-sample form)."""
    if cache is True or (cache == 'auto' and news_reps_cacheable(model)):
        return compute_scores_cached(model, corpus, batch_size)
    was_training = model.training
    model.eval()
    model.news_encoder.__dict__['_dedup_stats'] = [0, 0]
    scores = torch.zeros(corpus.num, device=corpus.device, dtype=torch.float32)
    for start in range(0, corpus.num, batch_size):
        idx = torch.arange(start, min(start + batch_size, corpus.num), device=corpus.device, dtype=torch.int32)
        batch = corpus.train_batch(idx)                       # candidate fields are [B, 1, ...] = the unsqueeze of util.py:43-48
        scores[start:start + idx.numel()] = model(*batch).squeeze(dim=1)
    model.train(was_training)
    rows = int(corpus.num * (corpus.H + 1))
    enc, of = model.news_encoder.__dict__.get('_dedup_stats', [0, 0])
    # CNE: history slots whose representation is the constant PAD representation are not encoded (news_encoders.cne_history_dedup)
    LAST_STATS.update(mode='per-sample', encoder_rows=rows - (of - enc), per_sample_rows=rows, pad_slots_skipped=of - enc)
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
