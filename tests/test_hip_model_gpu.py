"""End-to-end parity of the HIP path (nnr_amd.Model + Trainer, through libnnr_hip.so) against golden vectors captured
from the reference's own model.py on CPU.  Bar (BASELINE.json north_star): logits / loss within 1e-4 fp32."""
import numpy as np
import pytest
import torch

from golden_io import GoldenCase, ALL_CASES

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4          # the north-star bar
TIGHT = 2e-5              # what the fp32 kernels are expected to reach


def _build(case):
    from nnr_amd.model import Model
    cfg = case.config
    model = Model(cfg, case.word_table())
    case.load_into(model)
    model = model.cuda()
    model.train() if case.meta['mode'] == 'train' else model.eval()
    return model, cfg


def _supported(tag):
    return 'CNE_SUE' in tag or _have_all_encoders()


def _have_all_encoders():
    from nnr_amd import news_encoders, user_encoders
    return hasattr(news_encoders, 'MHSA') and hasattr(news_encoders, 'CNN') and hasattr(user_encoders, 'MHSA') and hasattr(user_encoders, 'ATT')


@pytest.mark.parametrize('tag', ALL_CASES)
def test_model_matches_reference_golden(tag):
    if not _supported(tag):
        pytest.skip('encoder pair not on the HIP path yet')
    from nnr_amd.trainer import Trainer
    from nnr_amd.model import negative_log_softmax
    case = GoldenCase(tag)
    model, cfg = _build(case)
    trainer = Trainer(model, cfg)
    steps = int(case.meta['adam_steps'])
    rec = {}
    ne = model.news_encoder
    if hasattr(ne, 'forward_pair'):                   # Model.forward drives CNE through the lock-step pair entry
        orig_pair = ne.forward_pair

        def recording_pair(c, h):
            a, b = orig_pair(c, h)
            rec['reps'] = [a.detach().cpu().numpy(), b.detach().cpu().numpy()]
            return a, b
        ne.forward_pair = recording_pair
    else:
        ne.register_forward_hook(lambda m, i, o: rec.setdefault('reps', []).append(o.detach().cpu().numpy()))
    ue = model.user_encoder
    orig_enc = ue.encode_user

    def recording_enc(*a):
        o = orig_enc(*a)
        rec['user'] = o.detach().cpu().numpy()
        return o
    ue.encode_user = recording_enc
    report = []
    for s in range(steps):
        batch = case.batch('cuda')
        trainer.flat.zero_grad()
        logits = model(*batch)
        loss = negative_log_softmax(logits)
        loss.backward()
        torch.cuda.synchronize()
        if s == 0:
            e = {k: float(np.abs(v - case.expect(n)).max()) for k, v, n in
                 (('cand_rep', rec['reps'][0], 'cand_rep'), ('hist_rep', rec['reps'][1], 'hist_rep'), ('user_rep', rec['user'], 'user_rep'))}
            report.append('stage max-abs-err: %s' % e)
            lg = logits.detach().cpu().numpy()
            err = float(np.abs(lg - case.expect('logits')).max())
            report.append('logits err %.3e  loss err %.3e' % (err, abs(float(loss) - float(case.expect('loss')))))
            print('\n'.join(report))
            assert max(e.values()) <= TIGHT * max(1.0, float(np.abs(case.expect('hist_rep')).max())), e
            assert err <= LOGIT_TOL and err <= TIGHT * max(1.0, float(np.abs(lg).max())), err
            assert abs(float(loss) - float(case.expect('loss'))) <= TIGHT
            # in-place input mutation is part of the reference's observable behaviour
            np.testing.assert_array_equal(batch[16].cpu().numpy(), case.expect('mutated_news_title_mask'))
            np.testing.assert_array_equal(batch[11].cpu().numpy(), case.expect('mutated_user_history_category_mask'))
            total = float(case.expect('grad_total_norm'))
            for k, p in model.named_parameters():
                if k.startswith('user_encoder.news_encoder.'):
                    continue
                exp, act = case.expect_grad(k, p.grad)
                scale = max(1e-3, float(case.expect('gradnorm/' + k)), 0.05 * total)
                assert float(np.abs(act - exp).max()) <= 5e-5 * scale, 'grad ' + k
                # per-tensor relative L2 (round-3 verdict: the bar above is relative to 5 % of the TOTAL norm, loose for small tensors):
                # every tensor that carries signal agrees with the reference's own gradient to 1e-3 (full arrays: the tiny-dim fixtures)
                nk = float(case.expect('gradnorm/' + k))
                if exp.size == p.numel() and nk > 1e-4 * total:
                    rel = float(np.linalg.norm((act - exp).astype(np.float64))) / nk
                    assert rel <= 1e-3, 'grad %s: relative L2 error %.3e' % (k, rel)
                gn = float(p.grad.double().norm())
                assert abs(gn - float(case.expect('gradnorm/' + k))) <= 5e-5 * scale, 'gradnorm ' + k
            assert abs(trainer.grad_total_norm() - total) <= 2e-5 * max(1.0, total)
        assert abs(float(loss) - float(case.expect('loss_step%d' % s))) <= 5e-5, 'loss at step %d' % s
        trainer.optimizer_step(1.0)
    torch.cuda.synchronize()
    lr = float(cfg.lr)
    for k, p in model.named_parameters():
        if k.startswith('user_encoder.news_encoder.'):
            continue
        exp, act = case.expect_param(steps, k, p)
        # Adam divides by sqrt(v): an element whose gradient is at the fp32 noise floor (or changes sign between two runs
        # because f32 atomics sum in a different order) legitimately moves by up to lr per step.  Hard bound on every
        # element, tight bound on the mean deviation (robust to those few elements).
        dlt = np.abs(act - exp)
        assert dlt.max(initial=0.0) <= steps * lr * 1.01 + 1e-4, 'param (hard bound) ' + k
        if float(case.expect('gradnorm/' + k)) >= 1e-2 * float(case.expect('grad_total_norm')):   # gradient well above the noise floor
            assert float(dlt.mean()) <= max(2e-5, 0.05 * steps * lr), 'param (mean deviation) ' + k


def test_missing_library_fails_loudly(monkeypatch):
    """The product path has no CPU fallback: CPU tensors are refused."""
    from nnr_amd import ops, _lib
    with pytest.raises(_lib.NnrHipError):
        ops.add_(torch.zeros(4), torch.zeros(4))


@pytest.mark.parametrize('tag', ['tiny_CNE_SUE_stable', 'full_CNE_SUE_g1p0_stable'])
def test_plugin_calls_equal_lockstep_path(tag):
    """The reference's plugin surface (news_encoder(...) then user_encoder(...), model.py:123-125) must give exactly what
    Model.forward's lock-step path gives (same kernels, only the recurrence launches are shared)."""
    case = GoldenCase(tag)
    model, cfg = _build(case)
    b = case.batch('cuda')
    logits = model(*b).detach()
    b = case.batch('cuda')
    (uid, ucat, usub, utt, utm, ute, uct, ucm, uce, uhm, ug, ucmask, ucidx, ncat, nsub, ntt, ntm, nte, nct, ncm, nce) = b
    cand = model.news_encoder(ntt, ntm, nte, nct, ncm, nce, ncat, nsub, None)
    user = model.user_encoder(utt, utm, ute, uct, ucm, uce, ucat, usub, uhm, ug, ucmask, ucidx, None, cand)
    plug = (user * cand).sum(dim=2)
    assert float((plug - logits).abs().max()) <= 1e-6
