#!/usr/bin/env python3
"""Generate tests/golden/eval_*.npz by running the REFERENCE's own evaluation path -- util.compute_scores (util.py:10-68) and
evaluate.scoring (evaluate.py:32-89) -- with the reference's own model.py on the dev split of a tiny synthetic MIND tree.

Runs only in the build container.  Accommodations (none touches the arithmetic): the import stand-ins of tools/ref_shims,
a SimpleNamespace config, a temp CWD, `Tensor.cuda()` / `torch.cuda.empty_cache()` as no-ops (no GPU here: the reference
moves every batch to the GPU), and for the CNE fixture torch.sort forced stable (= the reference's pinned torch 1.12.1, see
tools/make_goldens.py).  The fixture holds arrays only: the dev corpus tables, the model's state_dict, and what the
reference produced: one score per (impression, candidate) sample, the per-impression ranks it wrote to the result file, the
labels of its truth file, and (AUC, MRR, nDCG@5, nDCG@10).
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, os.path.join(ROOT, 'tools', 'ref_shims'))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
OUT = os.path.join(ROOT, 'tests', 'golden')

from make_corpus_goldens import write_tree          # noqa: E402
from make_goldens import stable_sort_patch          # noqa: E402
from oracle.nnr_oracle import default_config        # noqa: E402  (attribute bag only)


def run(tag, news, user, stable):
    rng = np.random.default_rng(21)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        work = os.path.join(tmp, 'work')
        os.makedirs(work)
        write_tree(os.path.join(tmp, 'MIND-tiny'), rng)
        os.chdir(work)
        orig_cuda, orig_empty = torch.Tensor.cuda, torch.cuda.empty_cache
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.cuda.empty_cache = lambda: None
        try:
            import MIND_corpus, util, model as ref_model        # /root/reference
            cfg = default_config(news_encoder=news, user_encoder=user, dataset='tiny', word_threshold=1, max_title_length=8,
                                 max_abstract_length=16, word_embedding_dim=16, hidden_dim=8, attention_dim=8, max_history_num=6,
                                 category_embedding_dim=4, subCategory_embedding_dim=4, negative_sample_num=2, head_num=2, head_dim=4,
                                 cnn_kernel_num=12, gcn_layer_num=2, dropout_rate=0.2, entity_embedding_dim=100, context_embedding_dim=100,
                                 no_self_connection=False, no_adjacent_normalization=False, gcn_normalization_type='symmetric',
                                 train_root='../MIND-tiny/train', dev_root='../MIND-tiny/dev', test_root='../MIND-tiny/test')
            torch.manual_seed(5)
            corpus = MIND_corpus.MIND_Corpus(cfg)
            m = ref_model.Model(cfg)
            m.initialize()
            with torch.no_grad():                               # larger weights: scores spread out (ranks are then robust to fp32 noise)
                for k, p in m.named_parameters():
                    if 'word_embedding' not in k:
                        p.mul_(2.0)
            os.makedirs('dev/ref')
            os.makedirs('dev/res')
            with open(os.path.join(cfg.dev_root, 'behaviors.tsv'), encoding='utf-8') as dev_f, open('dev/ref/truth-tiny.txt', 'w', encoding='utf-8') as truth_f:
                for dev_ID, line in enumerate(dev_f):           # config.py:158-163
                    impressions = line.split('\t')[4]
                    labels = [int(i[-1]) for i in impressions.strip().split(' ')]
                    truth_f.write(('' if dev_ID == 0 else '\n') + str(dev_ID + 1) + ' ' + str(labels).replace(' ', ''))
            # capture the per-sample scores compute_scores keeps in a local: wrap the model's forward
            scores = []
            fwd = m.forward
            m.forward = lambda *a: (lambda o: (scores.append(o.detach().clone().numpy().reshape(-1)), o)[1])(fwd(*a))
            auc, mrr, ndcg5, ndcg10 = util.compute_scores(m, corpus, 8, 'dev', 'dev/res/out.txt', 'tiny')
            m.forward = fwd
            ranks, labels, sizes = [], [], []
            with open('dev/res/out.txt') as f:
                for line in f:
                    r = json.loads(line.strip().split()[1])
                    ranks += r
                    sizes.append(len(r))
            with open('dev/ref/truth-tiny.txt') as f:
                for line in f:
                    labels += json.loads(line.strip().split()[1])
            out = dict(scores=np.concatenate(scores).astype(np.float32), ranks=np.array(ranks, dtype=np.int32), labels=np.array(labels, dtype=np.uint8),
                       sizes=np.array(sizes, dtype=np.int32), metrics=np.array([auc, mrr, ndcg5, ndcg10], dtype=np.float64),
                       dev_indices=np.array(corpus.dev_indices, dtype=np.int32),
                       category_num=np.int64(cfg.category_num), subCategory_num=np.int64(cfg.subCategory_num), vocabulary_size=np.int64(cfg.vocabulary_size),
                       user_num=np.int64(cfg.user_num), entity_size=np.int64(cfg.entity_size),
                       news_category=corpus.news_category, news_subCategory=corpus.news_subCategory, news_title_text=corpus.news_title_text,
                       news_title_mask=corpus.news_title_mask, news_title_entity=corpus.news_title_entity, news_abstract_text=corpus.news_abstract_text,
                       news_abstract_mask=corpus.news_abstract_mask, news_abstract_entity=corpus.news_abstract_entity,
                       beh_user=np.array([b[0] for b in corpus.dev_behaviors], dtype=np.int64),
                       beh_history=np.array([b[1] for b in corpus.dev_behaviors], dtype=np.int32),
                       beh_history_mask=np.array([b[2] for b in corpus.dev_behaviors], dtype=bool),
                       beh_candidate=np.array([b[3] for b in corpus.dev_behaviors], dtype=np.int32),
                       beh_line=np.array([b[4] for b in corpus.dev_behaviors], dtype=np.int32),
                       train_user_history_graph=corpus.dev_user_history_graph, train_user_history_category_mask=corpus.dev_user_history_category_mask,
                       train_user_history_category_indices=corpus.dev_user_history_category_indices)
            for k, v in m.state_dict().items():
                out['state/' + k] = v.numpy()
            cfgd = {k: v for k, v in vars(cfg).items() if isinstance(v, (int, float, str, bool))}
            out['cfg_keys'] = np.array(sorted(cfgd)).astype(str)
            out['cfg_vals'] = np.array([str(cfgd[k]) for k in sorted(cfgd)]).astype(str)
            out['cfg_types'] = np.array([type(cfgd[k]).__name__ for k in sorted(cfgd)]).astype(str)
            out['tie_order'] = np.array('stable' if stable else 'torch')
        finally:
            torch.Tensor.cuda, torch.cuda.empty_cache = orig_cuda, orig_empty
            os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, 'eval_%s.npz' % tag), **out)
    print(tag, 'samples', out['scores'].shape[0], 'impressions', out['sizes'].shape[0], 'metrics', out['metrics'], 'score range', out['scores'].min(), out['scores'].max())


def run_metrics():
    """evaluate.py's own scoring() and per-impression functions on 300 ragged impressions (2..300 candidates, 1..many clicks)."""
    import io
    import evaluate                                          # /root/reference/evaluate.py
    rng = np.random.default_rng(33)
    sizes = np.concatenate([[2, 2, 3, 300, 299, 17], rng.integers(2, 120, size=294)]).astype(np.int32)
    labels, ranks, per = [], [], []
    truth, sub = io.StringIO(), io.StringIO()
    for i, n in enumerate(sizes):
        y = np.zeros(n, dtype=np.uint8)
        y[rng.choice(n, size=int(rng.integers(1, max(2, min(n - 1, 6)))), replace=False)] = 1
        r = (rng.permutation(n) + 1).astype(np.int32)
        labels.append(y); ranks.append(r)
        truth.write(('' if i == 0 else '\n') + str(i + 1) + ' ' + str(y.tolist()).replace(' ', ''))
        sub.write(('' if i == 0 else '\n') + str(i + 1) + ' ' + str(r.tolist()).replace(' ', ''))
        ys, sc = y.astype('float32'), [1. / v for v in r]
        per.append([evaluate.roc_auc_score(ys, sc), evaluate.mrr_score(ys, sc), evaluate.ndcg_score(ys, sc, 5), evaluate.ndcg_score(ys, sc, 10)])
    truth.seek(0); sub.seek(0)
    means = evaluate.scoring(truth, sub)
    np.savez_compressed(os.path.join(OUT, 'eval_metrics_ragged.npz'), sizes=sizes, labels=np.concatenate(labels), ranks=np.concatenate(ranks),
                        per_impression=np.array(per, dtype=np.float64), metrics=np.array(means, dtype=np.float64))
    print('metrics_ragged', len(sizes), 'impressions', int(sizes.sum()), 'samples', means)


if __name__ == '__main__':
    torch.set_num_threads(4)
    run_metrics()
    run('tiny_MHSA_MHSA', 'MHSA', 'MHSA', False)
    run('tiny_CNN_ATT', 'CNN', 'ATT', False)
    with stable_sort_patch():
        run('tiny_CNE_SUE_stable', 'CNE', 'SUE', True)
