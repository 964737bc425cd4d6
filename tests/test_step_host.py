"""Host rules of the native step (nnr_amd/step.py) that need no GPU: which matrix path a step takes (round 6: bf16x3 by model and step size)."""
import torch

from nnr_amd import ops, step
from nnr_amd.config import make_config
from nnr_amd.model import Model


def _model(ne, ue):
    cfg = make_config(['--news_encoder=' + ne, '--user_encoder=' + ue, '--dataset=200k', '--batch_size=8'], corpus_sizes=dict(vocabulary_size=500))
    torch.manual_seed(0)
    return Model(cfg), cfg


def test_matrix_path_rule_by_model_and_step_size():
    cne, cfg = _model('CNE', 'SUE')
    mhsa, _ = _model('MHSA', 'MHSA')
    every = set(ops._BX3_CLASSES)
    per_sample = cfg.negative_sample_num + 1 + cfg.max_history_num            # 55 news-encoder sequences per impression
    on = ops.BX3[0]
    try:
        ops.BX3[0] = True
        # CNE + SUE: every class from 1 408 sequences per step on (per-GPU batch 32 = 1 760); batch 8 / 16 = 440 / 880 stay on the fp32 kernels
        assert step.bx3_classes(cne, 64 * per_sample) == every and step.bx3_classes(cne, 32 * per_sample) == every
        assert step.bx3_classes(cne, 16 * per_sample) == set() and step.bx3_classes(cne, 8 * per_sample) == set()
        assert step.bx3_classes(cne) == every                                      # size unknown: no size rule
        # MHSA news encoder: only the K >= 1024 data-gradient class, whatever the size
        assert step.bx3_classes(mhsa, 64 * per_sample) == {'dx'} and step.bx3_classes(mhsa, 8 * per_sample) == {'dx'}
        assert ops.bx3_class(300, 1200) == 'dx' and ops.bx3_class(1200, 300) == 'proj' and ops.bx3_class(900, 900) == 'sue' and ops.bx3_class(400, 400) == 'gate'
        # the scope restores what it found, also when the step turns the path off
        classes = ops._BX3_CLASSES
        batch = [None] * 21
        batch[15], batch[3] = torch.zeros(8, 5, 32, dtype=torch.int32), torch.zeros(8, 50, 32, dtype=torch.int32)
        assert step.step_sequences(batch) == 8 * 55
        with step.matrix_path(cne, batch):
            assert ops.BX3[0] is False
        assert ops.BX3[0] is True and ops._BX3_CLASSES is classes
        with step.matrix_path(mhsa, batch):
            assert ops.BX3[0] is True and ops._BX3_CLASSES == {'dx'}
        assert ops._BX3_CLASSES is classes
        ops.BX3[0] = False
        assert step.bx3_classes(cne, 64 * per_sample) == set() and step.bx3_classes(mhsa, 64 * per_sample) == set()
    finally:
        ops.BX3[0] = on
