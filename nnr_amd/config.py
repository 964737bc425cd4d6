"""Flag namespace of the hot path: the reference's `main.py` / `config.py` flag names and defaults (config.py:15-76)
plus its per-dataset override block (config.py:84-98).  Unlike the reference's Config it has no side effects: no GPU
assert, no dataset download, no directory creation (those belong to the control plane, out of scope)."""
import argparse
import json
import os
from types import SimpleNamespace

_FLAGS = [
    ('mode', str, 'train'), ('news_encoder', str, 'CNE'), ('user_encoder', str, 'SUE'), ('device_id', int, 0), ('seed', int, 0),
    ('config_file', str, ''), ('dataset', str, '200k'), ('tokenizer', str, 'MIND'), ('word_threshold', int, 3),
    ('max_title_length', int, 32), ('max_abstract_length', int, 128), ('negative_sample_num', int, 4), ('max_history_num', int, 50),
    ('epoch', int, 16), ('batch_size', int, 64), ('lr', float, 1e-4), ('weight_decay', float, 0.0), ('gradient_clip_norm', float, 4.0),
    ('world_size', int, 1), ('word_embedding_dim', int, 300), ('entity_embedding_dim', int, 100), ('context_embedding_dim', int, 100),
    ('cnn_method', str, 'naive'), ('cnn_kernel_num', int, 400), ('cnn_window_size', int, 3), ('attention_dim', int, 200),
    ('head_num', int, 20), ('head_dim', int, 20), ('user_embedding_dim', int, 50), ('category_embedding_dim', int, 50),
    ('subCategory_embedding_dim', int, 50), ('dropout_rate', float, 0.2), ('gcn_normalization_type', str, 'symmetric'), ('gcn_layer_num', int, 4), ('hidden_dim', int, 200),
    ('click_predictor', str, 'dot_product'),
]
_BOOL_FLAGS = ['no_self_connection', 'no_adjacent_normalization', 'no_gcn_residual', 'gcn_layer_norm']

NEWS_ENCODERS = ['CNE', 'CNN', 'MHSA']          # in scope; the reference lists 15
USER_ENCODERS = ['SUE', 'MHSA', 'ATT']          # in scope; the reference lists 11


def build_parser():
    p = argparse.ArgumentParser(description='NNR hot path on MI355X (flag names follow the reference config.py)')
    for name, typ, default in _FLAGS:
        p.add_argument('--' + name, type=typ, default=default, **({'choices': ['symmetric', 'asymmetric']} if name == 'gcn_normalization_type' else {}))
    for name in _BOOL_FLAGS:
        p.add_argument('--' + name, default=False, action='store_true')
    p.add_argument('--tie_order', type=str, default='stable', choices=['stable', 'torch'],
                   help="order of equal-length sequences in CNE's sort (see nnr_amd/news_encoders.py)")
    return p


def apply_dataset_overrides(cfg):
    """config.py:84-98 silently overwrites dropout_rate / gcn_layer_num / epoch per dataset."""
    if cfg.dataset == 'small':
        cfg.dropout_rate, cfg.gcn_layer_num = 0.25, 3
    elif cfg.dataset == '200k':
        cfg.dropout_rate, cfg.gcn_layer_num, cfg.epoch = 0.2, 4, 8
    else:
        cfg.dropout_rate, cfg.gcn_layer_num, cfg.epoch = 0.1, 4, 6
    return cfg


def make_config(argv=None, corpus_sizes=None, **over):
    """Parse flags (reference names), apply the dataset override block, then an optional --config_file JSON
    (config.py:100-110), then keyword overrides.  `corpus_sizes` injects vocabulary_size / category_num /
    subCategory_num / user_num / entity_size as MIND_Corpus.__init__ does (MIND_corpus.py:226-243)."""
    ns = build_parser().parse_args([] if argv is None else argv)
    cfg = SimpleNamespace(**vars(ns))
    apply_dataset_overrides(cfg)
    if cfg.config_file:
        if not os.path.exists(cfg.config_file):
            raise Exception('Config file does not exist : ' + cfg.config_file)
        with open(cfg.config_file, 'r', encoding='utf-8') as f:
            for k, v in json.load(f).items():
                if hasattr(cfg, k):
                    setattr(cfg, k, v)
    sizes = dict(vocabulary_size=60000, category_num=18, subCategory_num=285, user_num=1, entity_size=1)
    sizes.update(corpus_sizes or {})
    for k, v in sizes.items():
        setattr(cfg, k, v)
    for k, v in over.items():
        setattr(cfg, k, v)
    # config.py:111 and :116
    assert not (cfg.no_self_connection and not cfg.no_adjacent_normalization), 'Adjacent normalization of graph only can be set in case of self-connection'
    assert cfg.batch_size % cfg.world_size == 0, 'For multi-gpu training, batch size must be divisible by world size'
    return cfg
