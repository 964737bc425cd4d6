"""CPU oracle for the NNR training hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this module.  The shipped path (`nnr_amd/`) never does:
it calls hand-written HIP kernels through `libnnr_hip.so` and fails loudly when
that library is missing.

What this is: an independent restatement, in plain PyTorch CPU ops, of the
arithmetic the reference performs on the path
    trainer.py:81-120  ->  model.py:120-133  ->  newsEncoders.py / userEncoders.py / layers.py
written from the behavioural spec, not from the
reference's code structure:
  * the Bi-LSTM is an explicit time loop over a length-sorted active prefix
    (the reference uses nn.LSTM on a PackedSequence, newsEncoders.py:119-127);
  * torch_scatter's scatter_softmax / scatter_sum (userEncoders.py:88-89;
    third-party, torch_scatter==2.0.9 pinned in README.md:38, absent from
    /root/reference) are restated as one-hot cluster-membership contractions;
  * the GCN, attention pools and multi-head attention are written as einsums.
Parameter names / state-dict keys equal the reference's
so golden weights captured from the reference load unchanged.

Parity pin: `tests/test_oracle_golden.py` checks this module against golden
vectors under `tests/golden/` that `tools/make_goldens.py` produced by importing
the reference's own model.py in the build container (with a pure-torch stand-in
for torch_scatter, verified there against per-cluster loops).  The reference
ships no tests or known-answer vectors of its own.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- config
def default_config(**over):
    """Attribute bag with the flag names/defaults of config.py:15-76 that the path reads,
    plus the corpus-injected sizes (MIND_corpus.py:226-243)."""
    c = dict(
        news_encoder='CNE', user_encoder='SUE', click_predictor='dot_product',
        dataset='200k', tokenizer='MIND', word_threshold=3,
        max_title_length=32, max_abstract_length=128,
        negative_sample_num=4, max_history_num=50, batch_size=64,
        lr=1e-4, weight_decay=0.0, gradient_clip_norm=4.0, world_size=1, seed=0,
        word_embedding_dim=300, category_embedding_dim=50, subCategory_embedding_dim=50,
        cnn_method='naive', cnn_kernel_num=400, cnn_window_size=3,
        attention_dim=200, head_num=20, head_dim=20, hidden_dim=200,
        dropout_rate=0.2, gcn_layer_num=4, no_gcn_residual=False, gcn_layer_norm=False,
        vocabulary_size=1000, category_num=18, subCategory_num=285, user_num=1, entity_size=1,
    )
    c.update(over)
    return SimpleNamespace(**c)


# --------------------------------------------------------------------------- small pieces
def forced_dropout(x, p, training, keep=None):
    """F.dropout(x, p, training) -- or, when a parity test injects the keep-mask the HIP path's counter-based generator
    produced for this very site and call, the same arithmetic with THAT mask:  x * keep / (1 - p).  (Dropout masks are
    generator-specific, so a train-mode step of the two implementations can only be compared element for element with the
    mask shared; every dropout site of the path has such a hook: newsEncoders.py:53,117-118, userEncoders.py:80,91,171,
    layers.py:319-322.)"""
    if keep is None or not training or p <= 0.0:
        return F.dropout(x, p, training)
    return x * (keep.to(x.dtype).reshape(x.shape) * (1.0 / (1.0 - p)))


RELU_PROBE = None       # parity tests only: {'z': {}, 'u': {}, 'force': {site: bool mask} | None}, see probed_relu


def probed_relu(z, site, u=None):
    """torch.relu(z) -- plus, when a parity test set RELU_PROBE, a record of the pre-activation `z` (and of the affine layer's input
    `u`) of this user-encoder ReLU site, and optionally the ReLU's active set FORCED to a given mask (y = z * mask, derivative = mask).
    tests/test_hip_headline_gpu.py uses it to PROVE that a gradient deviation above the bar is a pre-activation that is zero to fp32
    rounding and landed on the other side of the kink in the HIP path: the oracle is re-run with the product's active set and must then
    agree at the strict bar.  Sites: 'gcn0'..'gcn{L-1}' (layers.py:285-292), 'affine' (userEncoders.py:91), 'mhsa_user' (userEncoders.py:171)."""
    pr = RELU_PROBE
    if pr is None:
        return torch.relu(z)
    pr['z'][site] = z.detach()
    if u is not None:
        pr['u'][site] = u.detach()
    force = (pr.get('force') or {}).get(site)
    if force is None:
        return torch.relu(z)
    return z * force.to(z.dtype).reshape(z.shape)


def masked_softmax(scores, mask, dim):
    """softmax after masked_fill(mask==0, -1e9)  (layers.py:143,171,199)."""
    if mask is not None:
        scores = torch.where(mask.bool(), scores, torch.full_like(scores, -1e9))
    return torch.softmax(scores, dim=dim)


class AdditivePool(nn.Module):
    """`Attention`, layers.py:151-175:  alpha = softmax(w2 . tanh(W1 x + b1)); out = sum alpha x."""

    def __init__(self, feature_dim, attention_dim):
        super().__init__()
        self.affine1 = nn.Linear(feature_dim, attention_dim, bias=True)
        self.affine2 = nn.Linear(attention_dim, 1, bias=False)

    def initialize(self):  # layers.py:157-160
        nn.init.xavier_uniform_(self.affine1.weight, gain=nn.init.calculate_gain('tanh'))
        nn.init.zeros_(self.affine1.bias)
        nn.init.xavier_uniform_(self.affine2.weight)

    def forward(self, x, mask=None):
        s = torch.tanh(self.affine1(x)) @ self.affine2.weight[0]      # [n, L]
        alpha = masked_softmax(s, mask, dim=1)
        return torch.einsum('nl,nlf->nf', alpha, x)


class CandidatePool(nn.Module):
    """`ScaledDotProduct_CandidateAttention`, layers.py:178-203."""

    def __init__(self, feature_dim, query_dim, attention_dim):
        super().__init__()
        self.K = nn.Linear(feature_dim, attention_dim, bias=False)
        self.Q = nn.Linear(query_dim, attention_dim, bias=True)
        self.scale = math.sqrt(float(attention_dim))

    def initialize(self):  # layers.py:185-188
        nn.init.xavier_uniform_(self.K.weight)
        nn.init.xavier_uniform_(self.Q.weight)
        nn.init.zeros_(self.Q.bias)

    def forward(self, x, query, mask=None):
        s = torch.einsum('nla,na->nl', self.K(x), self.Q(query)) / self.scale
        alpha = masked_softmax(s, mask, dim=1)
        return torch.einsum('nl,nlf->nf', alpha, x)


class MultiHead(nn.Module):
    """`MultiHeadAttention`, layers.py:102-148 (no output projection, key-side mask only)."""

    def __init__(self, h, d_model, len_q, len_k, d_k, d_v):
        super().__init__()
        self.h, self.d_k, self.d_v = h, d_k, d_v
        self.W_Q = nn.Linear(d_model, h * d_k)
        self.W_K = nn.Linear(d_model, h * d_k)
        self.W_V = nn.Linear(d_model, h * d_v)

    def initialize(self):  # layers.py:117-123
        for lin in (self.W_Q, self.W_K, self.W_V):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)

    def forward(self, xq, xk, xv, mask=None):
        n, lq, _ = xq.shape
        lk = xk.shape[1]
        q = self.W_Q(xq).reshape(n, lq, self.h, self.d_k)
        k = self.W_K(xk).reshape(n, lk, self.h, self.d_k)
        v = self.W_V(xv).reshape(n, lk, self.h, self.d_v)
        s = torch.einsum('nqhd,nkhd->nhqk', q, k) / math.sqrt(float(self.d_k))
        m = None if mask is None else mask.reshape(n, 1, 1, lk).expand_as(s)
        alpha = masked_softmax(s, m, dim=3)
        return torch.einsum('nhqk,nkhd->nqhd', alpha, v).reshape(n, lq, self.h * self.d_v)


class Conv1D(nn.Module):
    """`Conv1D` naive branch only (layers.py:13-14, 34-35)."""

    def __init__(self, cnn_method, in_channels, cnn_kernel_num, cnn_window_size):
        super().__init__()
        if cnn_method != 'naive':
            raise NotImplementedError('oracle covers cnn_method=naive only (SURVEY.md 2/#5)')
        self.conv = nn.Conv1d(in_channels, cnn_kernel_num, cnn_window_size, padding=(cnn_window_size - 1) // 2)

    def forward(self, x_ncl):
        return torch.relu(self.conv(x_ncl))


class BiLSTM(nn.Module):
    """One-layer bidirectional LSTM with nn.LSTM's parameter names, evaluated on the first
    `length[i]` tokens of each row exactly as pack_padded_sequence -> nn.LSTM -> pad_packed_sequence
    does (newsEncoders.py:119-127): outputs are zero past the length, c_n is the cell after the last
    valid step (forward) / after t=0 (reverse).  Gate order i,f,g,o (torch.nn.LSTM)."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.hidden_dim = hidden_dim
        for sfx in ('', '_reverse'):
            self.register_parameter('weight_ih_l0' + sfx, nn.Parameter(torch.empty(4 * hidden_dim, input_dim)))
            self.register_parameter('weight_hh_l0' + sfx, nn.Parameter(torch.empty(4 * hidden_dim, hidden_dim)))
            self.register_parameter('bias_ih_l0' + sfx, nn.Parameter(torch.empty(4 * hidden_dim)))
            self.register_parameter('bias_hh_l0' + sfx, nn.Parameter(torch.empty(4 * hidden_dim)))
        k = 1.0 / math.sqrt(hidden_dim)
        for p in self.parameters():
            nn.init.uniform_(p, -k, k)

    def _direction(self, xw, lens_sorted, w_hh, reverse):
        n, L, _ = xw.shape
        hd = self.hidden_dim
        h = xw.new_zeros(n, hd)
        c = xw.new_zeros(n, hd)
        active = [(int((lens_sorted > t).sum())) for t in range(L)]
        outs = [None] * L
        steps = range(L - 1, -1, -1) if reverse else range(L)
        for t in steps:
            nt = active[t]
            if nt == 0:
                outs[t] = xw.new_zeros(n, hd)
                continue
            z = xw[:nt, t] + h[:nt] @ w_hh.t()
            zi, zf, zg, zo = z.split(hd, dim=1)
            c_new = torch.sigmoid(zf) * c[:nt] + torch.sigmoid(zi) * torch.tanh(zg)
            h_new = torch.sigmoid(zo) * torch.tanh(c_new)
            c = torch.cat([c_new, c[nt:]], dim=0)
            h = torch.cat([h_new, h[nt:]], dim=0)
            outs[t] = torch.cat([h_new, xw.new_zeros(n - nt, hd)], dim=0)
        return torch.stack(outs, dim=1), c

    # 'loop': the explicit time loop below (independent of ATen's fused LSTM: the parity checker).
    # 'aten': ATen's own CPU LSTM on a PackedSequence -- the very code path the reference's nn.LSTM takes on the host
    #         (newsEncoders.py:119-127); same results (tests/test_oracle_golden.py runs both), several times faster, so this
    #         is what bench.py times as the CPU baseline.
    backend = 'loop'

    def _forward_aten(self, x, lengths):
        n, L, _ = x.shape
        packed = nn.utils.rnn.pack_padded_sequence(x, lengths.cpu(), batch_first=True, enforce_sorted=False)
        flat = [getattr(self, k + sfx) for sfx in ('', '_reverse') for k in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]
        zeros = x.new_zeros(2, n, self.hidden_dim)
        out, _, c_n = torch._VF.lstm(packed.data, packed.batch_sizes, (zeros, zeros), flat, True, 1, 0.0, self.training, True)
        H, _ = nn.utils.rnn.pad_packed_sequence(nn.utils.rnn.PackedSequence(out, packed.batch_sizes, packed.sorted_indices, packed.unsorted_indices),
                                                batch_first=True, total_length=L)
        c_n = c_n.index_select(1, packed.unsorted_indices)            # back to the caller's row order (nn.LSTM.permute_hidden)
        return H, torch.cat([c_n[0], c_n[1]], dim=1)

    def forward(self, x, lengths):
        """x [n, L, E] (values past length ignored), lengths [n] >= 1.
        Returns H [n, L, 2h] and c_n [n, 2h] = [c_fwd ; c_rev], both in the caller's row order."""
        if self.backend == 'aten':
            return self._forward_aten(x, lengths)
        order = torch.argsort(lengths, descending=True, stable=True)
        inv = torch.argsort(order)
        xs, ls = x[order], lengths[order]
        xw_f = xs @ self.weight_ih_l0.t() + (self.bias_ih_l0 + self.bias_hh_l0)
        xw_r = xs @ self.weight_ih_l0_reverse.t() + (self.bias_ih_l0_reverse + self.bias_hh_l0_reverse)
        hf, cf = self._direction(xw_f, ls, self.weight_hh_l0, reverse=False)
        hr, cr = self._direction(xw_r, ls, self.weight_hh_l0_reverse, reverse=True)
        return torch.cat([hf, hr], dim=2)[inv], torch.cat([cf, cr], dim=1)[inv]


def length_order(lengths, tie_order):
    """Permutation `torch.sort(lengths, descending=True)` yields at newsEncoders.py:112-115.
    The reference does not request a stable sort, so the order among equal lengths is whatever the
    installed torch's CPU sort does: stable in the pinned torch 1.12.1 (std::stable_sort), NOT stable in
    torch 2.10 (std::sort for n > 16).  Because the cross-selective gate pairs the two streams by sorted
    rank (CNE.forward below), the tie order is observable in the logits.
      tie_order='torch'  : call the installed torch exactly as the reference does (matches goldens
                           captured by running the reference under this container's torch);
      tie_order='stable' : stable descending order (pinned-version behaviour; what the device-side
                           planner of the HIP path computes without a host sync)."""
    lengths = lengths.long()
    if tie_order == 'stable':
        return torch.argsort(lengths, descending=True, stable=True)
    if tie_order == 'torch':
        return torch.sort(lengths.cpu(), descending=True)[1].to(lengths.device)
    raise ValueError(tie_order)


# --------------------------------------------------------------------------- news encoders
class NewsEncoder(nn.Module):
    """Tables + feature fusion, newsEncoders.py:11-54.  The word table is trainable with no padding_idx."""

    def __init__(self, config, word_table=None):
        super().__init__()
        self.word_embedding_dim = config.word_embedding_dim
        self.word_embedding = nn.Embedding(config.vocabulary_size, config.word_embedding_dim)
        if word_table is not None:          # the reference unpickles this from CWD (newsEncoders.py:16-17)
            with torch.no_grad():
                self.word_embedding.weight.copy_(word_table)
        self.category_embedding = nn.Embedding(config.category_num, config.category_embedding_dim)
        self.subCategory_embedding = nn.Embedding(config.subCategory_num, config.subCategory_embedding_dim)
        self.dropout_rate = config.dropout_rate
        self.auxiliary_loss = None

    # parity tests: {(site, call index): keep-mask}; sites 'title' / 'content' (newsEncoders.py:117-118, the title-only encoders
    # use 'title' for :163,193), 'cat' / 'sub' (:53), 'mid' (:165,197); call index 0 = the candidate call, 1 = the history call
    forced_keep = None
    _call_index = 0

    def drop(self, x, site=None):
        keep = None if self.forced_keep is None else self.forced_keep.get((site, self._call_index))
        return forced_dropout(x, self.dropout_rate, self.training, keep)

    def initialize(self):  # newsEncoders.py:24-27
        nn.init.uniform_(self.category_embedding.weight, -0.1, 0.1)
        nn.init.uniform_(self.subCategory_embedding.weight, -0.1, 0.1)
        with torch.no_grad():
            self.subCategory_embedding.weight[0].zero_()

    def feature_fusion(self, rep, category, subCategory):  # newsEncoders.py:50-54
        out = torch.cat([rep, self.drop(self.category_embedding(category), 'cat'),
                         self.drop(self.subCategory_embedding(subCategory), 'sub')], dim=2)
        self._call_index += 1          # (Model.forward resets it: candidate call, then history call)
        return out


class CNE(NewsEncoder):
    """newsEncoders.py:57-141."""

    def __init__(self, config, word_table=None):
        super().__init__(config, word_table)
        hd = config.hidden_dim
        self.T, self.C, self.hidden_dim = config.max_title_length, config.max_abstract_length, hd
        self.tie_order = getattr(config, 'tie_order', 'torch')
        self.news_embedding_dim = 4 * hd + config.category_embedding_dim + config.subCategory_embedding_dim
        self.title_lstm = BiLSTM(config.word_embedding_dim, hd)
        self.content_lstm = BiLSTM(config.word_embedding_dim, hd)
        self.title_H = nn.Linear(2 * hd, 2 * hd, bias=False)
        self.title_M = nn.Linear(2 * hd, 2 * hd, bias=True)
        self.content_H = nn.Linear(2 * hd, 2 * hd, bias=False)
        self.content_M = nn.Linear(2 * hd, 2 * hd, bias=True)
        self.title_self_attention = AdditivePool(2 * hd, config.attention_dim)
        self.content_self_attention = AdditivePool(2 * hd, config.attention_dim)
        self.title_cross_attention = CandidatePool(2 * hd, 2 * hd, config.attention_dim)
        self.content_cross_attention = CandidatePool(2 * hd, 2 * hd, config.attention_dim)

    def initialize(self):  # newsEncoders.py:79-100
        super().initialize()
        for lstm in (self.title_lstm, self.content_lstm):
            for p in lstm.parameters():
                if p.dim() >= 2:
                    nn.init.orthogonal_(p.data)
                else:
                    nn.init.zeros_(p.data)
        g = nn.init.calculate_gain('sigmoid')
        for lin in (self.title_H, self.title_M, self.content_H, self.content_M):
            nn.init.xavier_uniform_(lin.weight, gain=g)
        nn.init.zeros_(self.title_M.bias)
        nn.init.zeros_(self.content_M.bias)
        for a in (self.title_self_attention, self.content_self_attention,
                  self.title_cross_attention, self.content_cross_attention):
            a.initialize()

    def forward(self, title_text, title_mask, title_entity, content_text, content_mask, content_entity,
                category, subCategory, user_embedding):
        B, N = title_text.shape[:2]
        n = B * N
        tmask = title_mask.view(n, self.T)
        cmask = content_mask.view(n, self.C)
        tmask[:, 0] = 1            # in place on the caller's tensor, newsEncoders.py:108-109
        cmask[:, 0] = 1
        tlen = tmask.sum(dim=1).long()
        clen = cmask.sum(dim=1).long()
        xt = self.drop(self.word_embedding(title_text), 'title').view(n, self.T, -1)
        xc = self.drop(self.word_embedding(content_text), 'content').view(n, self.C, -1)
        Ht, mt = self.title_lstm(xt, tlen)
        Hc, mc = self.content_lstm(xc, clen)
        # cross-selective gate, newsEncoders.py:128-131 (padded rows stay zero because H is zero there).
        # Reference quirk that parity must keep: the gate is formed while BOTH streams are still in their
        # own length-sorted order (:112-115), so the title at title-rank r is gated by the content memory of
        # the news at content-rank r (and vice versa) -- generally a different news; see length_order().
        order_t = length_order(tlen, self.tie_order)
        order_c = length_order(clen, self.tie_order)
        rank_t = torch.argsort(order_t)
        rank_c = torch.argsort(order_c)
        mc_partner = mc[order_c[rank_t]]
        mt_partner = mt[order_t[rank_c]]
        Ht = Ht * torch.sigmoid(self.title_H(Ht) + self.title_M(mc_partner)[:, None, :])
        Hc = Hc * torch.sigmoid(self.content_H(Hc) + self.content_M(mt_partner)[:, None, :])
        t_self = self.title_self_attention(Ht, tmask)
        c_self = self.content_self_attention(Hc, cmask)
        t_cross = self.title_cross_attention(Ht, c_self, tmask)      # query = the other stream, :136-137
        c_cross = self.content_cross_attention(Hc, t_self, cmask)
        rep = torch.cat([t_self + t_cross, c_self + c_cross], dim=1).view(B, N, 4 * self.hidden_dim)
        return self.feature_fusion(rep, category, subCategory)


class CNN(NewsEncoder):
    """newsEncoders.py:144-170."""

    def __init__(self, config, word_table=None):
        super().__init__(config, word_table)
        self.T = config.max_title_length
        self.conv = Conv1D(config.cnn_method, config.word_embedding_dim, config.cnn_kernel_num, config.cnn_window_size)
        self.attention = AdditivePool(config.cnn_kernel_num, config.attention_dim)
        self.news_embedding_dim = config.cnn_kernel_num + config.category_embedding_dim + config.subCategory_embedding_dim

    def initialize(self):
        super().initialize()
        self.attention.initialize()

    def forward(self, title_text, title_mask, title_entity, content_text, content_mask, content_entity,
                category, subCategory, user_embedding):
        B, N = title_text.shape[:2]
        mask = title_mask.view(B * N, self.T)
        w = self.drop(self.word_embedding(title_text), 'title').view(B * N, self.T, -1)
        c = self.drop(self.conv(w.transpose(1, 2)).transpose(1, 2), 'mid')
        rep = self.attention(c, mask).view(B, N, -1)
        return self.feature_fusion(rep, category, subCategory)


class MHSA(NewsEncoder):
    """newsEncoders.py:173-200."""

    def __init__(self, config, word_table=None):
        super().__init__(config, word_table)
        self.T = config.max_title_length
        self.feature_dim = config.head_num * config.head_dim
        self.multiheadAttention = MultiHead(config.head_num, config.word_embedding_dim, self.T, self.T,
                                            config.head_dim, config.head_dim)
        self.attention = AdditivePool(self.feature_dim, config.attention_dim)
        self.news_embedding_dim = self.feature_dim + config.category_embedding_dim + config.subCategory_embedding_dim

    def initialize(self):
        super().initialize()
        self.multiheadAttention.initialize()
        self.attention.initialize()

    def forward(self, title_text, title_mask, title_entity, content_text, content_mask, content_entity,
                category, subCategory, user_embedding):
        B, N = title_text.shape[:2]
        mask = title_mask.view(B * N, self.T)
        w = self.drop(self.word_embedding(title_text), 'title').view(B * N, self.T, -1)
        c = self.drop(self.multiheadAttention(w, w, w, mask), 'mid')
        rep = self.attention(c, mask).view(B, N, self.feature_dim)
        return self.feature_fusion(rep, category, subCategory)


# --------------------------------------------------------------------------- user encoders
class UserEncoder(nn.Module):
    """userEncoders.py:12-39; owns a reference to the shared news-encoder instance (:16)."""

    def __init__(self, news_encoder, config):
        super().__init__()
        self.news_embedding_dim = news_encoder.news_embedding_dim
        self.news_encoder = news_encoder
        self.auxiliary_loss = None

    def encode_history(self, a):
        return self.news_encoder(a['user_title_text'], a['user_title_mask'], a['user_title_entity'],
                                 a['user_content_text'], a['user_content_mask'], a['user_content_entity'],
                                 a['user_category'], a['user_subCategory'], a['user_embedding'])

    _ARGS = ('user_title_text', 'user_title_mask', 'user_title_entity', 'user_content_text', 'user_content_mask',
             'user_content_entity', 'user_category', 'user_subCategory', 'user_history_mask', 'user_history_graph',
             'user_history_category_mask', 'user_history_category_indices', 'user_embedding',
             'candidate_news_representation')

    def forward(self, *args):
        return self.encode_user(dict(zip(self._ARGS, args)))


class GCNLayer(nn.Module):
    def __init__(self, dim, residual, layer_norm):
        super().__init__()
        self.residual, self.layer_norm = residual, layer_norm
        self.W = nn.Linear(dim, dim, bias=True)
        if layer_norm:
            self.layer_normalization = nn.LayerNorm([dim])

    def initialize(self):  # layers.py:276-278
        nn.init.xavier_uniform_(self.W.weight, gain=nn.init.calculate_gain('relu'))
        nn.init.zeros_(self.W.bias)

    def forward(self, x, graph):  # layers.py:285-292
        u = torch.einsum('bij,bjd->bid', graph, x)
        y = self.W(u)
        if self.layer_norm:
            y = self.layer_normalization(y)
        y = probed_relu(y, getattr(self, '_site', 'gcn'), u)
        return y + x if self.residual else y


class GCN(nn.Module):
    """layers.py:294-323: dropout (p = rate/2) between layers, none after the last."""

    def __init__(self, dim, num_layers, dropout, residual, layer_norm):
        super().__init__()
        self.p = dropout
        self.gcn_layers = nn.ModuleList([GCNLayer(dim, residual, layer_norm) for _ in range(num_layers)])

    def initialize(self):
        for l in self.gcn_layers:
            l.initialize()

    forced_keep = None          # parity tests: {layer index: keep-mask [B, G, D]} (layers.py:319-322)

    def forward(self, x, graph):
        for i, layer in enumerate(self.gcn_layers):
            layer._site = 'gcn%d' % i
            x = layer(x, graph)
            if i + 1 < len(self.gcn_layers):
                x = forced_dropout(x, self.p, self.training, None if self.forced_keep is None else self.forced_keep.get(i))
        return x


class SUE(UserEncoder):
    """userEncoders.py:42-98."""

    def __init__(self, news_encoder, config):
        super().__init__(news_encoder, config)
        D = self.news_embedding_dim
        self.attention_dim = max(config.attention_dim, D // 4)
        self.proxy_node_embedding = nn.Parameter(torch.zeros(config.category_num, D))
        self.gcn = GCN(D, config.gcn_layer_num, config.dropout_rate / 2, not config.no_gcn_residual, config.gcn_layer_norm)
        self.intraCluster_K = nn.Linear(D, self.attention_dim, bias=False)
        self.intraCluster_Q = nn.Linear(D, self.attention_dim, bias=True)
        self.clusterFeatureAffine = nn.Linear(D, D, bias=True)
        self.interClusterAttention = CandidatePool(D, D, self.attention_dim)
        self.p = config.dropout_rate
        self.forced_keep = None
        self.cluster_num = config.category_num + 1      # +1: the padding cluster
        self.H = config.max_history_num

    def initialize(self):  # userEncoders.py:58-66
        self.gcn.initialize()
        nn.init.zeros_(self.proxy_node_embedding)
        nn.init.xavier_uniform_(self.intraCluster_K.weight)
        nn.init.xavier_uniform_(self.intraCluster_Q.weight)
        nn.init.zeros_(self.intraCluster_Q.bias)
        nn.init.xavier_uniform_(self.clusterFeatureAffine.weight, gain=nn.init.calculate_gain('relu'))
        nn.init.zeros_(self.clusterFeatureAffine.bias)
        self.interClusterAttention.initialize()

    def encode_user(self, a):
        cand = a['candidate_news_representation']                     # [B, N, D]
        B, N, D = cand.shape
        cmask = a['user_history_category_mask']
        cmask[:, -1] = 1                                              # in place, userEncoders.py:73
        idx = a['user_history_category_indices']                      # [B, H] int64 in [0, K]
        hist = self.encode_history(a)                                 # [B, H, D]
        # the reference applies dropout_ to the batch-expanded proxy tensor (one mask per sample), :80
        fk = self.forced_keep or {}         # parity tests: 'proxy' [B, K, D] (userEncoders.py:80), 'affine' [B, N, C, D] (:91)
        proxy = forced_dropout(self.proxy_node_embedding.unsqueeze(0).expand(B, -1, -1), self.p, self.training, fk.get('proxy'))
        x0 = torch.cat([hist, proxy], dim=1)                          # [B, G, D]
        g = (self.gcn(x0, a['user_history_graph']) + x0)[:, :self.H]  # [B, H, D]
        # intra-cluster attention == scatter_softmax / scatter_sum over the cluster index (userEncoders.py:85-89)
        kf = self.intraCluster_K(g)                                   # [B, H, A]
        qc = self.intraCluster_Q(cand)                                # [B, N, A]
        s = torch.einsum('bja,bna->bnj', kf, qc) / math.sqrt(float(self.attention_dim))
        member = F.one_hot(idx, self.cluster_num).to(s.dtype)         # [B, H, C]
        neg = torch.finfo(s.dtype).min
        smax = torch.where(member.bool().unsqueeze(1), s.unsqueeze(3), s.new_full((), neg)).amax(dim=2)  # [B,N,C]
        pick = idx.unsqueeze(1).expand(-1, N, -1)                     # cluster id of item j, per candidate
        e = torch.exp(s - torch.gather(smax, 2, pick))
        denom = torch.einsum('bnj,bjc->bnc', e, member)               # per-cluster sums
        alpha = e / torch.gather(denom, 2, pick)
        feat = torch.einsum('bnj,bjc,bjd->bncd', alpha, member, g)    # empty clusters -> 0
        feat = forced_dropout(probed_relu(self.clusterFeatureAffine(feat), 'affine', feat) + feat, self.p, self.training, fk.get('affine'))
        C = self.cluster_num
        out = self.interClusterAttention(feat.reshape(B * N, C, D), cand.reshape(B * N, D),
                                         cmask.unsqueeze(1).expand(-1, N, -1).reshape(B * N, C))
        return out.view(B, N, D)


class MHSAUser(UserEncoder):
    """userEncoders.py:151-173.  Note the hard-wired F.dropout default p=0.5 at :171."""

    def __init__(self, news_encoder, config):
        super().__init__(news_encoder, config)
        D = self.news_embedding_dim
        self.multiheadAttention = MultiHead(config.head_num, D, config.max_history_num, config.max_history_num,
                                            config.head_dim, config.head_dim)
        self.affine = nn.Linear(config.head_num * config.head_dim, D, bias=True)
        self.attention = AdditivePool(D, config.attention_dim)

    def initialize(self):
        self.multiheadAttention.initialize()
        nn.init.xavier_uniform_(self.affine.weight, gain=nn.init.calculate_gain('relu'))
        nn.init.zeros_(self.affine.bias)
        self.attention.initialize()

    def encode_user(self, a):
        N = a['candidate_news_representation'].shape[1]
        hist = self.encode_history(a)
        h = self.multiheadAttention(hist, hist, hist, a['user_history_mask'])
        forced = getattr(self, 'forced_dropout_keep', None)
        if forced is not None and self.training:
            # parity tests: the keep-mask of the HIP path's counter-based generator for this call, injected so that the
            # train-mode arithmetic of userEncoders.py:171 (p = 0.5, scale 2) can be compared element for element
            y = self.affine(h)
            if RELU_PROBE is not None:
                # (parity tests' kink proof: relu(2 m y) == 2 m relu(y) bit for bit for m in {0, 1}; the probe sees the pre-dropout y)
                h = probed_relu(y, 'mhsa_user', h) * (forced.to(y.dtype).view_as(y) * 2.0)
            else:
                h = torch.relu(y * forced.to(y.dtype).view_as(y) * 2.0)
        else:
            h = torch.relu(F.dropout(self.affine(h), 0.5, self.training))
        return self.attention(h).unsqueeze(1).repeat(1, N, 1)        # unmasked pool, :172


class ATT(UserEncoder):
    """userEncoders.py:176-191: unmasked additive pool over the 50 history slots."""

    def __init__(self, news_encoder, config):
        super().__init__(news_encoder, config)
        self.attention = AdditivePool(self.news_embedding_dim, config.attention_dim)

    def initialize(self):
        self.attention.initialize()

    def encode_user(self, a):
        N = a['candidate_news_representation'].shape[1]
        return self.attention(self.encode_history(a)).unsqueeze(1).expand(-1, N, -1)


# --------------------------------------------------------------------------- model + step
_NEWS = {'CNE': CNE, 'CNN': CNN, 'MHSA': MHSA}
_USER = {'SUE': SUE, 'MHSA': MHSAUser, 'ATT': ATT}


class Model(nn.Module):
    """model.py:10-133 restricted to the in-scope encoders and the dot-product click predictor."""

    def __init__(self, config, word_table=None):
        super().__init__()
        if config.news_encoder not in _NEWS or config.user_encoder not in _USER:
            raise NotImplementedError('oracle covers CNE/CNN/MHSA x SUE/MHSA/ATT (SURVEY.md section 8a)')
        if config.click_predictor != 'dot_product':
            raise NotImplementedError('oracle covers click_predictor=dot_product (model.py:126-127)')
        self.news_encoder = _NEWS[config.news_encoder](config, word_table)
        self.user_encoder = _USER[config.user_encoder](self.news_encoder, config)
        self.model_name = config.news_encoder + '-' + config.user_encoder
        self.news_embedding_dim = self.news_encoder.news_embedding_dim
        self.use_user_embedding = False
        self.click_predictor = config.click_predictor

    def initialize(self):
        self.news_encoder.initialize()
        self.user_encoder.initialize()

    def forward(self, user_ID, user_category, user_subCategory, user_title_text, user_title_mask, user_title_entity,
                user_content_text, user_content_mask, user_content_entity, user_history_mask, user_history_graph,
                user_history_category_mask, user_history_category_indices,
                news_category, news_subCategory, news_title_text, news_title_mask, news_title_entity,
                news_content_text, news_content_mask, news_content_entity):
        self.news_encoder._call_index = 0
        cand = self.news_encoder(news_title_text, news_title_mask, news_title_entity, news_content_text,
                                 news_content_mask, news_content_entity, news_category, news_subCategory, None)
        user = self.user_encoder(user_title_text, user_title_mask, user_title_entity, user_content_text,
                                 user_content_mask, user_content_entity, user_category, user_subCategory,
                                 user_history_mask, user_history_graph, user_history_category_mask,
                                 user_history_category_indices, None, cand)
        return (user * cand).sum(dim=2)


def negative_log_softmax(logits):
    """trainer.py:64-66."""
    return -(torch.log_softmax(logits, dim=1)[:, 0]).mean()


def train_step(model, optimizer, batch, gradient_clip_norm=4.0):
    """One optimizer step as trainer.py:105-120: forward, loss, zero_grad, backward, clip, Adam."""
    logits = model(*batch)
    loss = negative_log_softmax(logits)
    optimizer.zero_grad()
    loss.backward()
    if gradient_clip_norm > 0:
        nn.utils.clip_grad_norm_(model.parameters(), gradient_clip_norm)
    optimizer.step()
    return logits.detach(), float(loss.detach())


def make_optimizer(model, config):
    """trainer.py:27: Adam over every trainable parameter (word table included)."""
    return torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=config.lr,
                            weight_decay=config.weight_decay)
