"""Load a golden case produced by tools/make_goldens.py (test infrastructure)."""
import os
from types import SimpleNamespace

import numpy as np
import torch

from golden_weights import make_state
from nnr_amd.synth import BATCH_FIELDS

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DROPOUT_CASES = ['drop_tiny_CNE_SUE_stable', 'drop_tiny_MHSA_MHSA', 'drop_tiny_CNN_ATT']      # round 3: reference in train mode, dropout ON
ALL_CASES = ['tiny_CNE_SUE', 'tiny_CNE_SUE_scaled', 'tiny_MHSA_MHSA', 'tiny_CNN_ATT',
             'tiny_CNE_SUE_stable', 'full_CNE_SUE_g1p0', 'full_CNE_SUE_g1p6', 'full_CNE_SUE_g1p0_stable',
             'full_MHSA_MHSA_g1p0', 'full_CNN_ATT_g1p0',
             # round 2 (tools/make_goldens.py extra): --gcn_layer_norm, with / without residual; hidden_dim 48 and 112
             'tiny_CNE_SUE_ln_stable', 'tiny_CNE_SUE_ln_nores_stable', 'tiny_CNE_SUE_h48_stable', 'tiny_CNE_SUE_h112_stable']


def _parse(v):
    if v in ('True', 'False'):
        return v == 'True'
    for cast in (int, float):
        try:
            return cast(v)
        except ValueError:
            pass
    return v


class GoldenCase:
    def __init__(self, tag):
        self.tag = tag
        self.z = np.load(os.path.join(GOLDEN_DIR, tag + '.npz'))
        self.meta = {k: _parse(v) for k, v in zip(self.z['meta_keys'], self.z['meta_vals'])}
        self.config = SimpleNamespace(**self.meta)
        self.full_arrays = bool(self.meta['full_arrays'])

    def batch(self, device='cpu'):
        return [torch.from_numpy(self.z['in/' + k].copy()).to(device) for k in BATCH_FIELDS]

    def param_names(self):
        pre = 'gradnorm/'
        return [k[len(pre):] for k in self.z.files if k.startswith(pre)]

    def initial_state(self, shapes):
        """{name: np.ndarray} initial parameters (stored for reference-initialised cases, regenerated otherwise)."""
        if self.meta['gain'] < 0:
            return {k: self.z['param0/' + k] for k in shapes}
        return make_state(shapes, self.meta['seed'], self.meta['gain'])

    def word_table(self):
        t = self.z['word_table']
        return None if t.size == 0 else torch.from_numpy(t.copy())

    def load_into(self, model):
        shapes = {k: tuple(p.shape) for k, p in model.named_parameters()}
        assert sorted(shapes) == sorted(self.param_names()), 'parameter names differ from the reference'
        st = self.initial_state(shapes)
        with torch.no_grad():
            for k, p in model.named_parameters():
                assert tuple(st[k].shape) == tuple(p.shape), k
                p.copy_(torch.from_numpy(np.ascontiguousarray(st[k])).to(p.device))

    def dropout_calls(self):
        """[(p, keep-mask)] in the order the reference called F.dropout (fixtures of `tools/make_goldens.py dropout`)."""
        out = []
        for i, p in enumerate(self.z['drop_p']):
            shape = tuple(int(v) for v in self.z['drop_shape/%d' % i])
            n = int(np.prod(shape))
            out.append((float(p), torch.from_numpy(np.unpackbits(self.z['drop_bits/%d' % i])[:n].astype(bool).reshape(shape))))
        return out

    def inject_dropout(self, model):
        """Hand the recorded masks to the oracle model's dropout sites.  The call order of the reference (model.py:123-125):
        candidate news-encoder call, history news-encoder call (each: word rows [title, content for CNE], [mid dropout for MHSA /
        CNN], category rows, subCategory rows), then the user encoder (SUE: proxy nodes, GCN layers 0..L-2, cluster affine;
        MHSA: the p = 0.5 site).  Also checks each site's p against the reference's."""
        calls = self.dropout_calls()
        cfg, it = self.config, iter(calls)
        p = float(cfg.dropout_rate)
        news = {}
        per_call = {'CNE': ('title', 'content', 'cat', 'sub'), 'MHSA': ('title', 'mid', 'cat', 'sub'), 'CNN': ('title', 'mid', 'cat', 'sub')}[cfg.news_encoder]
        for call in (0, 1):
            for site in per_call:
                pp, keep = next(it)
                assert abs(pp - p) < 1e-12, (site, pp)
                news[(site, call)] = keep
        model.news_encoder.forced_keep = news
        if cfg.user_encoder == 'SUE':
            pp, proxy = next(it)
            assert abs(pp - p) < 1e-12
            gcn = {}
            for l in range(cfg.gcn_layer_num - 1):
                pp, keep = next(it)
                assert abs(pp - p / 2) < 1e-12, pp               # GCN(dropout=dropout_rate / 2), userEncoders.py:47 / layers.py:301
                gcn[l] = keep
            pp, aff = next(it)
            assert abs(pp - p) < 1e-12
            model.user_encoder.forced_keep = {'proxy': proxy, 'affine': aff}
            model.user_encoder.gcn.forced_keep = gcn
        elif cfg.user_encoder == 'MHSA':
            pp, keep = next(it)
            assert pp == 0.5                                     # F.dropout's default, userEncoders.py:171
            model.user_encoder.forced_dropout_keep = keep
        assert next(it, None) is None, 'the reference called dropout more often than the oracle has sites'
        return len(calls)

    def expect(self, key):
        return self.z[key]

    def expect_param(self, step, name, actual):
        """Compare a parameter after `step` Adam steps (full array or its first 64 elements)."""
        e = self.z['param%d/%s' % (step, name)]
        a = actual.detach().cpu().numpy()
        return e, (a if self.full_arrays else a.reshape(-1)[:64])

    def expect_grad(self, name, actual):
        e = self.z['grad/' + name]
        a = actual.detach().cpu().numpy()
        return e, (a if self.full_arrays else a.reshape(-1)[:64])
