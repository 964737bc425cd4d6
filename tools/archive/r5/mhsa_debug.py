import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from nnr_amd.config import make_config
from nnr_amd.synth import SynthSpec, SynthCorpus, BATCH_FIELDS, to_torch
from nnr_amd import news_encoders as NE
from test_hip_edge_gpu import _models
cfg = make_config(['--news_encoder=MHSA', '--user_encoder=MHSA'], corpus_sizes=dict(vocabulary_size=800), dropout_rate=0.0, batch_size=4)
model, ref = _models(cfg, seed=11)
corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size, news_pool=400, seed=13))
for variant in ('plain', 'full_hist', 'full_cand', 'gap'):
    b = corpus.batch(4, np.random.default_rng(14))
    if variant == 'full_hist':
        b['user_title_mask'][1, 7, :] = False
    if variant == 'full_cand':
        b['news_title_mask'][2, 3, :] = False
    if variant == 'gap':
        b['user_title_mask'][0, 2, :6] = True
        b['user_title_mask'][0, 2, 2] = False
    b = {k: np.ascontiguousarray(v) for k, v in b.items()}
    with torch.no_grad():
        rl = ref(*to_torch(b))
        NE._MHSA_PACKED = True
        lp = model(*to_torch(b, 'cuda')).cpu()
        NE._MHSA_PACKED = False
        ld = model(*to_torch(b, 'cuda')).cpu()
        d = dict(zip(BATCH_FIELDS, to_torch(b, 'cuda')))
        c = dict(zip(BATCH_FIELDS, to_torch(b)))
        args = lambda x, pre: (x[pre + '_title_text'], x[pre + '_title_mask'], x[pre + '_title_entity'], x[pre + '_content_text'], x[pre + '_content_mask'], x[pre + '_content_entity'], x[pre + '_category'], x[pre + '_subCategory'], None)
        ro = ref.news_encoder(*args(c, 'user'))
        ref.news_encoder._call_index = 0
        NE._MHSA_PACKED = True
        rp = model.news_encoder(*args(d, 'user')).cpu()
        NE._MHSA_PACKED = False
        rd = model.news_encoder(*args(d, 'user')).cpu()
    print(variant, 'logits packed-oracle %.3e dense-oracle %.3e | history reps packed-oracle %.3e dense-oracle %.3e' % (
        float((lp - rl).abs().max()), float((ld - rl).abs().max()), float((rp - ro).abs().max()), float((rd - ro).abs().max())))
    bad = ((rp - ro).abs().amax(-1) > 1e-4).nonzero().tolist()
    print('   packed reps differ at (sample, history slot):', bad[:10], ' dense:', ((rd - ro).abs().amax(-1) > 1e-4).nonzero().tolist()[:10])
