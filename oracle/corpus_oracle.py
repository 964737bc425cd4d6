"""CPU restatement (numpy) of the data side of the hot path -- TEST INFRASTRUCTURE ONLY (checker for nnr_amd.corpus and
csrc/corpus.hip; never imported by the product path).

* history_graph : the per-behaviour user-history graph, cluster mask and cluster indices the reference pre-computes in
                  MIND_Corpus.preprocess step 6 (MIND_corpus.py:162-221).
* train_batch   : MIND_Train_Dataset.__getitem__ + default collate (MIND_dataset.py:70-76): the 21 batch tensors from the
                  corpus tables, for a list of behaviour indices.
Pinned against the reference itself: tests/golden/corpus_*.npz hold what /root/reference's MIND_corpus.py / MIND_dataset.py
produced on a tiny MIND tree (tools/make_corpus_goldens.py); tests/test_corpus_oracle.py checks this file against them
bit for bit."""
import numpy as np


def history_graph(cats, n_hist, max_history_num, category_num, norm='symmetric', self_connection=True):
    """cats: category ids of ONE behaviour line's history in file order (any length >= n_hist); n_hist: its true length.
    Returns (graph [G,G] float32, category_mask [category_num+1] bool, category_indices [max_history_num] int64),
    G = max_history_num + category_num.  MIND_corpus.py:179-216."""
    H, K = max_history_num, category_num
    G = H + K
    if norm == 'none_noself':                          # --no_self_connection: zero diagonal, never normalised (config.py:56,111)
        self_connection = False
    graph = np.identity(G, dtype=np.float32) if self_connection else np.zeros([G, G], dtype=np.float32)   # :180-183
    mask = np.zeros(K + 1, dtype=bool)                                                                      # :184
    indices = np.full([H], K, dtype=np.int64)                                                               # :185
    if n_hist > 0:                                                                                          # :186
        offset = max(0, n_hist - H)                                                                         # :188  (the LAST H items)
        n = min(n_hist, H)                                                                                  # :189
        for i in range(n):
            ci = int(cats[i + offset])
            mask[ci] = True
            indices[i] = ci
            graph[i, H + ci] = 1                                                                            # :194-195  news <-> its proxy node
            graph[H + ci, i] = 1
            for j in range(i + 1, n):
                cj = int(cats[j + offset])
                if ci == cj:
                    graph[i, j] = 1                                                                         # :199-200  same-category clique
                    graph[j, i] = 1
                else:
                    graph[H + ci, H + cj] = 1                                                               # :202-203  proxy <-> proxy
                    graph[H + cj, H + ci] = 1
        if norm == 'asymmetric':                                                                            # :205-209  D^-1 A  (float32 throughout)
            d = (1 / graph.sum(axis=1, keepdims=False)).astype(np.float32)
            graph = d[:, None] * graph
        elif norm == 'symmetric':                                                                           # :210-214  D^-1/2 A D^-1/2
            d = np.sqrt(1 / graph.sum(axis=1, keepdims=False)).astype(np.float32)
            graph = (d[:, None] * graph) * d[None, :]
    return graph, mask, indices


def train_batch(c, idx):
    """c: dict of corpus arrays (news_* tables, beh_user / beh_history / beh_history_mask / beh_line, train_samples,
    train_user_history_graph / _category_mask / _category_indices); idx: behaviour indices.  Returns the 21 arrays in the
    order of MIND_dataset.py:75-76 (= the argument order of trainer.py:105-106), stacked like the default collate."""
    idx = np.asarray(idx)
    hist = c['beh_history'][idx]                      # [B, H]   news indices
    samp = c['train_samples'][idx]                    # [B, 1+K]
    line = c['beh_line'][idx]
    news = lambda key, sel: c[key][sel]               # numpy fancy indexing, as the reference
    return [c['beh_user'][idx],
            news('news_category', hist), news('news_subCategory', hist), news('news_title_text', hist), news('news_title_mask', hist),
            news('news_title_entity', hist), news('news_abstract_text', hist), news('news_abstract_mask', hist), news('news_abstract_entity', hist),
            c['beh_history_mask'][idx], c['train_user_history_graph'][line], c['train_user_history_category_mask'][line],
            c['train_user_history_category_indices'][line],
            news('news_category', samp), news('news_subCategory', samp), news('news_title_text', samp), news('news_title_mask', samp),
            news('news_title_entity', samp), news('news_abstract_text', samp), news('news_abstract_mask', samp), news('news_abstract_entity', samp)]
