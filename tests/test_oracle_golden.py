"""Pins the CPU oracle (oracle/nnr_oracle.py) against golden vectors captured from the reference's own
model.py (tools/make_goldens.py).  Tolerances are the fp32 noise floor of the path (3e-8 at
init), far inside the 1e-4 logits/loss bar of BASELINE.json."""
import numpy as np
import pytest
import torch

from golden_io import GoldenCase, ALL_CASES, DROPOUT_CASES
from oracle import nnr_oracle as O


def _run(case, steps, dropout=False):
    cfg = case.config
    torch.manual_seed(0)
    model = O.Model(cfg, case.word_table())
    case.load_into(model)
    model.train() if case.meta['mode'] == 'train' else model.eval()
    if dropout:
        case.inject_dropout(model)
    opt = O.make_optimizer(model, cfg)
    res = []
    for _ in range(steps):
        batch = case.batch()
        logits = model(*batch)
        loss = O.negative_log_softmax(logits)
        opt.zero_grad()
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.gradient_clip_norm)
        opt.step()
        res.append((logits.detach(), float(loss), grads, float(norm), batch))
    return model, res


def check_params_after_adam(case, model, steps, tight=5e-5):
    """Adam's update is lr*m/(sqrt(v)+eps): where the gradient is at the fp32 noise floor its SIGN is noise
    and the parameter legitimately moves by +-lr per step.  Elements with a well-resolved first-step gradient
    must match tightly; the rest only within steps*lr."""
    lr = float(case.config.lr)
    for k, p in model.named_parameters():
        e, a = case.expect_param(steps, k, p)
        g = np.abs(case.expect('grad/' + k)).reshape(e.shape)
        # resolved = well above the tensor's own scale AND above the fp32 noise floor of the whole backward pass (a tensor whose
        # largest gradient is 1e-6 of the total norm is noise everywhere: Adam's g / (|g| + eps) turns that noise into +-lr)
        resolved = (g > 0.05 * max(float(g.max()), 1e-30)) & (g > 1e-4 * float(case.expect('grad_total_norm')))
        d = np.abs(a - e)
        assert d[resolved].max(initial=0.0) <= tight, k
        assert d.max(initial=0.0) <= steps * lr * 1.01 + tight, k


@pytest.fixture(params=['loop', 'aten'])
def lstm_backend(request, monkeypatch):
    """Both restatements of the Bi-LSTM are pinned: the explicit time loop (the parity checker) and ATen's own packed-sequence
    LSTM (what the reference's nn.LSTM runs on the host; bench.py's CPU baseline times this one)."""
    monkeypatch.setattr(O.BiLSTM, 'backend', request.param)
    return request.param


@pytest.mark.parametrize('tag', ALL_CASES)
def test_oracle_matches_reference_golden(tag, lstm_backend):
    if lstm_backend == 'aten' and 'CNE' not in tag:
        pytest.skip('no LSTM in this model')
    case = GoldenCase(tag)
    steps = int(case.meta['adam_steps'])
    model, res = _run(case, steps)
    logits, loss, grads, norm, batch = res[0]
    np.testing.assert_allclose(logits.numpy(), case.expect('logits'), rtol=0, atol=2e-6)
    assert abs(loss - float(case.expect('loss'))) < 2e-6
    assert abs(norm - float(case.expect('grad_total_norm'))) < 1e-5 * max(1.0, norm)
    for k, g in grads.items():
        e, a = case.expect_grad(k, g)
        # fp32 noise of a gradient entry scales with the whole backward pass, not with this tensor's norm
        scale = max(1e-3, float(case.expect('gradnorm/' + k)), 0.05 * float(case.expect('grad_total_norm')))
        assert np.abs(a - e).max() <= 2e-5 * scale, k
        gn = float(np.linalg.norm(g.numpy().astype(np.float64)))
        assert abs(gn - float(case.expect('gradnorm/' + k))) <= 1e-5 * scale, k
    # in-place input mutation is observable in the reference (newsEncoders.py:108-109, userEncoders.py:73)
    np.testing.assert_array_equal(batch[16].numpy(), case.expect('mutated_news_title_mask'))
    np.testing.assert_array_equal(batch[11].numpy(), case.expect('mutated_user_history_category_mask'))
    for s in range(steps):
        assert abs(res[s][1] - float(case.expect('loss_step%d' % s))) < 5e-6
    check_params_after_adam(case, model, steps)


@pytest.mark.parametrize('tag', DROPOUT_CASES)
def test_oracle_dropout_on_matches_reference(tag):
    """Train mode, dropout ON: the reference ran with recorded keep-masks (tools/make_goldens.py dropout); replayed at the
    oracle's sites the whole step must agree -- pins the position, rate, mask shape and 1/(1-p) scale of every dropout site
    (newsEncoders.py:53,117-118,163,165,193,196; userEncoders.py:80,91,171; layers.py:319-322) to the reference itself."""
    case = GoldenCase(tag)
    model, res = _run(case, 1, dropout=True)
    logits, loss, grads, norm, batch = res[0]
    np.testing.assert_allclose(logits.numpy(), case.expect('logits'), rtol=0, atol=5e-6)
    assert abs(loss - float(case.expect('loss'))) < 5e-6
    assert abs(norm - float(case.expect('grad_total_norm'))) < 1e-5 * max(1.0, norm)
    for k, g in grads.items():
        e, a = case.expect_grad(k, g)
        scale = max(1e-3, float(case.expect('gradnorm/' + k)), 0.05 * float(case.expect('grad_total_norm')))
        assert np.abs(a - e).max() <= 2e-5 * scale, k
    check_params_after_adam(case, model, 1)


def test_forced_dropout_is_F_dropout_with_the_mask_shared():
    """The hook's arithmetic is F.dropout's: with the mask torch itself drew, the outputs are bit-identical."""
    x = torch.randn(7, 5, 11)
    for p in (0.1, 0.2, 0.25, 0.5):
        torch.manual_seed(3)
        y = torch.nn.functional.dropout(x, p, True)
        keep = y != 0
        assert torch.equal(O.forced_dropout(x, p, True, keep), y)
    assert torch.equal(O.forced_dropout(x, 0.2, False, keep), x)


def test_oracle_state_dict_keys_match_reference():
    """Checkpoint compatibility (trainer.py:183): every reference parameter name exists, incl. the aliases."""
    case = GoldenCase('tiny_CNE_SUE')
    model = O.Model(case.config, case.word_table())
    sd = model.state_dict()
    for k in case.param_names():
        assert k in sd
        if k.startswith('news_encoder.'):
            assert 'user_encoder.' + k in sd
