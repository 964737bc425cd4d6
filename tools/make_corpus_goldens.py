#!/usr/bin/env python3
"""Generate tests/golden/corpus_*.npz by running the REFERENCE's own MIND_corpus.py / MIND_dataset.py on a tiny
synthetic MIND tree (news.tsv / behaviors.tsv / *.vec written to a temp directory).

Runs only in the build container (needs /root/reference); the fixtures hold arrays only: the corpus tables the reference
built, its per-behaviour user-history graphs / cluster masks / cluster indices (MIND_corpus.py:162-221), the behaviour
lists, and one DataLoader batch (MIND_dataset.py:70-76).  Accommodations: nltk / torchtext import
stand-ins (tools/ref_shims), a SimpleNamespace config, a temp CWD for the reference's json/pkl side files.

Usage:  python tools/make_corpus_goldens.py
"""
import json
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, os.path.join(ROOT, 'tools', 'ref_shims'))
sys.path.insert(0, REF)
OUT = os.path.join(ROOT, 'tests', 'golden')

CATS = ['news', 'sports', 'finance', 'travel', 'video', 'health', 'autos']
WORDS = ('the of and a in to is it for on with as at by from that this was are be or an new said year one two people time '
         'game team market stock city world state season first last day home school police court health car road win loss').split()


def write_tree(root, rng, n_news=60, max_hist=70):
    news_ids = ['N%d' % (1000 + i) for i in range(n_news)]
    cat_of = {}
    lines = []
    for i, nid in enumerate(news_ids):
        cat = CATS[int(rng.integers(0, len(CATS) - (1 if i < n_news // 2 else 0)))]
        cat_of[nid] = cat
        sub = cat + '_' + str(int(rng.integers(0, 3)))
        tl = int(rng.integers(1, 12))
        al = int(rng.integers(0, 40))
        title = ' '.join(rng.choice(WORDS, tl)) + (' 2021' if i % 7 == 0 else '')
        abstract = ' '.join(rng.choice(WORDS, al)) if al else ''
        ents = json.dumps([{'WikidataId': 'Q%d' % (i % 9), 'OccurrenceOffsets': [0]}]) if i % 3 == 0 else '[]'
        lines.append('\t'.join([nid, cat, sub, title, abstract, 'http://x/' + nid, ents, '[]']))
    splits = {'train': lines[:45], 'dev': lines[30:55], 'test': lines[40:]}
    users = ['U%d' % i for i in range(12)]
    for split, nl in splits.items():
        d = os.path.join(root, split)
        os.makedirs(d)
        with open(os.path.join(d, 'news.tsv'), 'w', encoding='utf-8') as f:
            f.write('\n'.join(nl) + '\n')
        ids = [l.split('\t')[0] for l in nl]
        with open(os.path.join(d, 'behaviors.tsv'), 'w', encoding='utf-8') as f:
            for b in range(14 if split == 'train' else 5):
                hl = [0, 1, 3, 7, 12, 49, 50, 51, max_hist][b % 9]
                hist = ' '.join(rng.choice(ids, hl)) if hl else ''
                k = int(rng.integers(2, 9))
                imp = rng.choice(ids, k, replace=False)
                labels = [1] + [0] * (k - 1) if b % 4 else [1, 1] + [0] * (k - 2)
                imps = ' '.join('%s-%d' % (n, l) for n, l in zip(imp, labels))
                f.write('\t'.join([str(b + 1), users[int(rng.integers(0, len(users)))], '11/11/2019 9:05:58 AM', hist, imps]) + '\n')
        for vec in ('entity_embedding.vec', 'context_embedding.vec'):
            with open(os.path.join(d, vec), 'w', encoding='utf-8') as f:
                for q in range(9):
                    f.write('Q%d\t' % q + '\t'.join('%.4f' % v for v in rng.normal(size=100)) + '\n')
    return cat_of


def run(tag, max_history_num, norm):
    rng = np.random.default_rng(7)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        work = os.path.join(tmp, 'work')
        os.makedirs(work)
        write_tree(os.path.join(tmp, 'MIND-tiny'), rng)
        os.chdir(work)
        try:
            import MIND_corpus                      # /root/reference/MIND_corpus.py
            import MIND_dataset                     # /root/reference/MIND_dataset.py
            cfg = SimpleNamespace(dataset='tiny', tokenizer='MIND', word_threshold=1, max_title_length=8, max_abstract_length=16,
                                  word_embedding_dim=50, entity_embedding_dim=100, context_embedding_dim=100,
                                  max_history_num=max_history_num, negative_sample_num=4, no_self_connection=(norm == 'none_noself'),
                                  no_adjacent_normalization=norm.startswith('none'), gcn_normalization_type=(norm if not norm.startswith('none') else 'symmetric'),
                                  train_root='../MIND-tiny/train', dev_root='../MIND-tiny/dev', test_root='../MIND-tiny/test')
            torch.manual_seed(0)
            corpus = MIND_corpus.MIND_Corpus(cfg)
            ds = MIND_dataset.MIND_Train_Dataset(corpus)
            MIND_dataset.randint = np.random.RandomState(3).randint          # the module-level `randint` the sampler calls
            ds.negative_sampling()
            from torch.utils.data import DataLoader
            idx = [0, 3, 5, len(ds) - 1, 2]
            batch = next(iter(DataLoader([ds[i] for i in idx], batch_size=len(idx), shuffle=False)))
            out = dict(
                category_num=np.int64(cfg.category_num), max_history_num=np.int64(max_history_num), norm=np.array(norm),
                news_category=corpus.news_category, news_subCategory=corpus.news_subCategory,
                news_title_text=corpus.news_title_text, news_title_mask=corpus.news_title_mask, news_title_entity=corpus.news_title_entity,
                news_abstract_text=corpus.news_abstract_text, news_abstract_mask=corpus.news_abstract_mask,
                news_abstract_entity=corpus.news_abstract_entity,
                train_user_history_graph=corpus.train_user_history_graph, train_user_history_category_mask=corpus.train_user_history_category_mask,
                train_user_history_category_indices=corpus.train_user_history_category_indices,
                beh_user=np.array([b[0] for b in corpus.train_behaviors], dtype=np.int64),
                beh_history=np.array([b[1] for b in corpus.train_behaviors], dtype=np.int32),
                beh_history_mask=np.array([b[2] for b in corpus.train_behaviors], dtype=bool),
                beh_line=np.array([b[5] for b in corpus.train_behaviors], dtype=np.int32),
                train_samples=np.array(ds.train_samples, dtype=np.int32), batch_index=np.array(idx, dtype=np.int32))
            # the raw history of every behaviours.tsv line (category ids in file order: what the graph builder consumes)
            with open('news_ID-tiny.json') as f:
                nid = json.load(f)
            hist_cat, hist_len = [], []
            with open(os.path.join(cfg.train_root, 'behaviors.tsv'), encoding='utf-8') as f:
                for line in f:
                    h = line.split('\t')[3].strip()
                    ids = [nid[x] for x in h.split(' ')] if h else []
                    hist_len.append(len(ids))
                    hist_cat.append([int(corpus.news_category[i]) for i in ids] + [-1] * (80 - len(ids)))
            out['line_history_category'] = np.array(hist_cat, dtype=np.int32)
            out['line_history_len'] = np.array(hist_len, dtype=np.int32)
            for k, t in enumerate(batch):
                out['batch_%02d' % k] = t.numpy() if torch.is_tensor(t) else np.asarray(t)
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, 'corpus_%s.npz' % tag), **out)
    print(tag, 'news', out['news_category'].shape[0], 'behaviours', out['beh_user'].shape[0], 'graph', out['train_user_history_graph'].shape,
          'batch fields', sum(k.startswith('batch_') for k in out))


if __name__ == '__main__':
    run('tiny_h50_sym', 50, 'symmetric')
    run('tiny_h8_asym', 8, 'asymmetric')
    run('tiny_h8_none', 8, 'none')
    run('tiny_h8_noself', 8, 'none_noself')     # --no_self_connection (config.py:56; requires --no_adjacent_normalization, config.py:111)
