"""Keep-masks of the HIP path's counter-based dropout generator, recomputed per dropout site for ONE model call and injected
into the CPU oracle (oracle.nnr_oracle.forced_dropout), so that a TRAIN-mode, dropout-ON step of the two implementations can
be compared element for element.  Dropout masks are generator-specific (the reference draws from torch's Philox stream, the
kernels from a hash of (seed, element index), csrc/common.h:nnr_keep), so this is the only way to pin the configuration the
bench measures: a wrong seed / offset / index formula / 1/(1-p) scale at ANY site of the forward gather, the weight-gradient
loaders or the scatter epilogues shows up as a logits / gradient mismatch.

The masks are produced on the GPU by the library's own generator (nnr_dropout of a vector of ones = keep(seed, flat index)),
laid out per site exactly as the kernels index them:
  CNE   word rows     (newsEncoders.py:117-118)  seed+1 / seed+2, flat index = packed token row * E + column
  all   category/sub  (newsEncoders.py:53)       seed+3 / seed+4, flat over [news of the call, 50]
  MHSA / CNN  words   (newsEncoders.py:163,193)  seed+1 flat over [n*L, E];  mid dropout (:165,197) seed+2 flat over [n*L, F]
  SUE   proxy nodes   (userEncoders.py:80)       seed+1, flat over [B, K, D]  (one mask per sample)
  SUE   GCN layer l   (layers.py:319-322)        seed+10+l, p/2, flat over [B, G, D], not after the last layer
  SUE   cluster affine (userEncoders.py:91)      seed+2, flat over [B*N*(K+1), D]
  MHSA user           (userEncoders.py:171)      seed (of the user encoder's call), p = 0.5, flat over [B*H, D]
The per-call seeds mirror NewsEncoder._next_seed / UserEncoder._next_seed for the NEXT call(s) of the given model."""
import torch


def flat_keep(numel, p, seed, device='cuda'):
    from nnr_amd import ops
    if p <= 0.0:
        return torch.ones(numel, dtype=torch.bool, device=device)
    return ops.dropout(torch.ones(numel, device=device), p, seed) > 0


def _news_seed(ne, k):
    return (ne._seed_base + 104729 * (ne._calls + k)) & 0x7FFFFFFF


def _user_seed(ue, k=1):
    return (ue._seed_base + 15485863 * (ue._calls + k)) & 0x7FFFFFFF


def _packed_to_dense(plan, keep_rows, n, L, E):
    """keep_rows [cap, E] (packed token rows) -> [n, L, E] in the caller's row order; positions past a sequence's length do not
    exist on the packed path (the oracle ignores them too: packed LSTM) and are set to keep."""
    dev = keep_rows.device
    rank = plan.rank.long()                                    # sorted position of original row i
    lens = plan.len.long()
    off = plan.off.long()[:L]
    t = torch.arange(L, device=dev)
    rows = off[None, :] + rank[:, None]                        # packed row of (i, t)
    valid = t[None, :] < lens[:, None]
    rows = torch.where(valid, rows, torch.zeros_like(rows))
    out = keep_rows[rows.reshape(-1)].view(n, L, E)
    out |= ~valid[:, :, None]
    return out


def cne_masks(model, batch_dev, union=True):
    """{(site, call): bool tensor on the CPU} for the next forward of a CNE model on `batch_dev` (dict of device tensors)."""
    from nnr_amd import ops
    ne = model.news_encoder
    p = ne.dropout_rate
    E = ne.word_embedding_dim
    B, N = batch_dev['news_title_text'].shape[:2]
    Hn = batch_dev['user_title_text'].shape[1]
    n0, n1 = B * N, B * Hn
    out = {}
    calls = [(0, 'news', n0), (1, 'user', n1)]
    if union:
        seed = _news_seed(ne, 1)
        for site, key, L, so in (('title', 'title', ne.max_title_length, 1), ('content', 'content', ne.max_content_length, 2)):
            m0 = batch_dev['news_%s_mask' % key].clone().view(n0, L)
            m1 = batch_dev['user_%s_mask' % key].clone().view(n1, L)
            plan = ops.SeqPlan(m0, None, None, m1, None)
            keep = flat_keep(plan.cap * E, p, seed + so).view(plan.cap, E)
            dense = _packed_to_dense(plan, keep, n0 + n1, L, E)
            out[(site, 0)] = dense[:n0].cpu()
            out[(site, 1)] = dense[n0:].cpu()
        for site, so in (('cat', 3), ('sub', 4)):
            k = flat_keep((n0 + n1) * 50, p, seed + so).view(n0 + n1, 50)
            out[(site, 0)] = k[:n0].cpu()
            out[(site, 1)] = k[n0:].cpu()
        return out
    for call, pre, n in calls:
        seed = _news_seed(ne, 1 + call)
        for site, key, L, so in (('title', 'title', ne.max_title_length, 1), ('content', 'content', ne.max_content_length, 2)):
            m = batch_dev['%s_%s_mask' % (pre, key)].clone().view(n, L)
            plan = ops.SeqPlan(m, None)
            keep = flat_keep(plan.cap * E, p, seed + so).view(plan.cap, E)
            out[(site, call)] = _packed_to_dense(plan, keep, n, L, E).cpu()
        for site, so in (('cat', 3), ('sub', 4)):
            out[(site, call)] = flat_keep(n * 50, p, seed + so).view(n, 50).cpu()
    return out


def dense_news_masks(model, batch_dev):
    """MHSA / CNN news encoders: two calls (candidate, history), each with its own seed."""
    from nnr_amd import news_encoders as NE, ops
    ne = model.news_encoder
    p = ne.dropout_rate
    E = ne.word_embedding_dim
    F = ne.news_embedding_dim - 100
    L = ne.max_sentence_length
    B, N = batch_dev['news_title_text'].shape[:2]
    Hn = batch_dev['user_title_text'].shape[1]
    out = {}
    # round 5: the MHSA news encoder runs over PACKED token rows (functional.MhsaPack): its word-row and attention-output masks are
    # indexed by (packed row, column), as CNE's are
    packed = type(ne).__name__ == 'MHSA' and NE.mhsa_packed(ne, batch_dev['news_title_text'])
    for call, n, key in ((0, B * N, 'news'), (1, B * Hn, 'user')):
        seed = _news_seed(ne, 1 + call)
        if packed:
            plan = ops.SeqPlan(ops.mask_cover(batch_dev[key + '_title_mask'].reshape(n, L).contiguous()), None)
            out[('title', call)] = _packed_to_dense(plan, flat_keep(plan.cap * E, p, seed + 1).view(plan.cap, E), n, L, E).cpu()
            out[('mid', call)] = _packed_to_dense(plan, flat_keep(plan.cap * F, p, seed + 2).view(plan.cap, F), n, L, F).cpu()
            out[('cat', call)] = flat_keep(n * 50, p, seed + 3).view(n, 50).cpu()
            out[('sub', call)] = flat_keep(n * 50, p, seed + 4).view(n, 50).cpu()
            continue
        out[('title', call)] = flat_keep(n * L * E, p, seed + 1).view(n, L, E).cpu()
        out[('mid', call)] = flat_keep(n * L * F, p, seed + 2).view(n, L, F).cpu()
        out[('cat', call)] = flat_keep(n * 50, p, seed + 3).view(n, 50).cpu()
        out[('sub', call)] = flat_keep(n * 50, p, seed + 4).view(n, 50).cpu()
    return out


def sue_masks(model, batch_dev):
    """(SUE forced_keep dict, GCN forced_keep dict) for the next call of model.user_encoder."""
    ue = model.user_encoder
    p = ue.dropout_rate
    D = ue.news_embedding_dim
    B, N = batch_dev['news_title_text'].shape[:2]
    Hn = batch_dev['user_title_text'].shape[1]
    Kc = ue.proxy_node_embedding.shape[0]
    G, Cn = Hn + Kc, Kc + 1
    seed = _user_seed(ue)
    sue = {'proxy': flat_keep(B * Kc * D, p, seed + 1).view(B, Kc, D).cpu(),
           'affine': flat_keep(B * N * Cn * D, p, seed + 2).view(B, N, Cn, D).cpu()}
    Lg = ue.gcn.num_layers
    gcn = {l: flat_keep(B * G * D, ue.gcn.dropout_rate, seed + 10 + l).view(B, G, D).cpu() for l in range(Lg - 1)}
    return sue, gcn


def mhsa_user_mask(model, batch_dev):
    ue = model.user_encoder
    B, Hn = batch_dev['user_title_text'].shape[:2]
    return flat_keep(B * Hn * ue.news_embedding_dim, 0.5, _user_seed(ue)).view(B, Hn, ue.news_embedding_dim).cpu()


def inject(model, ref, batch_dev, union=None):
    """Compute every dropout site's keep-mask for the NEXT training call of `model` on `batch_dev` and hand them to the oracle
    model `ref`.  Returns the mean keep rate per site (tests assert it is 1 - p)."""
    from nnr_amd import news_encoders as NE
    name = type(model.news_encoder).__name__
    if name == 'CNE':
        news = cne_masks(model, batch_dev, NE._CNE_UNION if union is None else union)
    else:
        news = dense_news_masks(model, batch_dev)
    ref.news_encoder.forced_keep = news
    rates = {'%s/%d' % k: float(v.float().mean()) for k, v in news.items()}
    uname = type(model.user_encoder).__name__
    if uname == 'SUE':
        sue, gcn = sue_masks(model, batch_dev)
        ref.user_encoder.forced_keep = sue
        ref.user_encoder.gcn.forced_keep = gcn
        rates.update({'sue/' + k: float(v.float().mean()) for k, v in sue.items()})
        rates.update({'gcn/%d' % k: float(v.float().mean()) for k, v in gcn.items()})
    elif uname == 'MHSA':
        ref.user_encoder.forced_dropout_keep = mhsa_user_mask(model, batch_dev)
    return rates
