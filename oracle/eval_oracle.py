"""CPU restatement (numpy, float64 like the reference) of the evaluation tail -- TEST INFRASTRUCTURE ONLY.

* ranks_from_scores   : util.py:50-59 -- inside every impression candidates are ranked by descending score; Python's
                        list.sort(reverse=True) is stable, so equal scores keep their original order (earlier candidate = better rank).
* impression_metrics  : evaluate.py:8-29,76-81 on y_score = 1 / rank: roc_auc_score, MRR, nDCG@5, nDCG@10.
* scoring             : evaluate.py:32-89 -- the means over impressions.
Pinned by tests/golden/eval_metrics_ragged.npz (evaluate.py's own functions on 300 ragged impressions) and by the
end-to-end eval fixtures (util.compute_scores with the reference's model on a tiny MIND dev split)."""
import numpy as np


def ranks_from_scores(scores, sizes):
    scores = np.asarray(scores)
    out = np.zeros(scores.shape[0], dtype=np.int32)
    o = 0
    for n in sizes:
        s = scores[o:o + n]
        order = sorted(range(n), key=lambda i: s[i], reverse=True)       # stable, like util.py:55
        for j, i in enumerate(order):
            out[o + i] = j + 1
        o += n
    return out


def impression_metrics(labels, ranks):
    """-> (auc, mrr, ndcg5, ndcg10) for ONE impression."""
    y = np.asarray(labels, dtype=np.float64)
    r = np.asarray(ranks, dtype=np.int64)
    order = np.argsort(r)                                     # descending 1/rank = ascending rank (a permutation: no ties)
    ys = y[order]
    P, N = ys.sum(), len(ys) - ys.sum()
    # roc_auc_score with distinct scores = fraction of (positive, negative) pairs ordered correctly
    neg_below = np.cumsum((1 - ys)[::-1])[::-1] - (1 - ys)    # negatives ranked after each position
    auc = float((ys * neg_below).sum() / (P * N))
    mrr = float((ys / (np.arange(len(ys)) + 1)).sum() / P)    # evaluate.py:24-28

    def ndcg(k):                                              # evaluate.py:8-21 (binary gains: 2**y - 1 = y)
        dcg = (ys[:k] / np.log2(np.arange(len(ys[:k])) + 2)).sum()
        best_y = np.sort(y)[::-1][:k]
        best = (best_y / np.log2(np.arange(len(best_y)) + 2)).sum()
        return float(dcg / best)
    return auc, mrr, ndcg(5), ndcg(10)


def scoring(labels, ranks, sizes):
    per, o = [], 0
    for n in sizes:
        per.append(impression_metrics(labels[o:o + n], ranks[o:o + n]))
        o += n
    per = np.array(per, dtype=np.float64)
    return per, per.mean(axis=0)
