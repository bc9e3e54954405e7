"""GPU: BASELINE config 1 (SNGAN-MNIST 32x32, LeNet, K=32, N=8, B=16) — one full training step on the HIP path
against the golden produced by the REFERENCE modules (tools/make_golden.py::gen_step)."""
import types

import pytest
import torch

from tests.test_oracle_golden import cfg1_setup
from tests.util import rel_err
from warpedganspace_amd.sngan import SNGANWrapper
from warpedganspace_amd.support_sets import SupportSets
from warpedganspace_amd.trainer import TrainStep

pytestmark = pytest.mark.gpu


def test_cfg1_step_vs_reference_golden(dev, golden):
    g = golden('step_cfg1')
    Gw, c, R, mag, (K, N, B, d) = cfg1_setup()
    S = SupportSets(K, N, d, learn_gammas=True, gamma=c['gamma'])
    S.load_state_dict(c['sd'])
    params = types.SimpleNamespace(reconstructor_lr=1e-4, support_set_lr=1e-4, min_shift_magnitude=0.25,
                                   max_shift_magnitude=0.45, lambda_cls=1.0, lambda_reg=0.25, z_truncation=None,
                                   shift_in_w_space=False)
    eng = TrainStep(SNGANWrapper(Gw).to(dev).eval(), S.to(dev).train(), R.to(dev).train(), params, B, dev, seed=0)
    st = eng.step(c['z'].to(dev), c['idx'].to(dev), mag.to(dev)).tolist()
    assert abs(st[0] - float(g['step_ce'])) < 1e-4 and abs(st[1] - float(g['step_l1'])) < 1e-4
    assert abs(st[2] - float(g['step_loss'])) < 1e-4 and abs(st[3] - float(g['step_acc'])) < 1e-6
    assert torch.equal(eng.argmax.cpu(), torch.from_numpy(g['step_argmax']))            # path-index argmax bit-exact
    gb = eng.bucket.gview
    rows = torch.unique(c['idx'])
    e_s = rel_err(gb[id(eng.S.SUPPORT_SETS)][rows.to(dev)][:, ::4], g['step_dS_rows'])
    e_g = rel_err(gb[id(eng.S.LOGGAMMA)], g['step_dloggamma'])
    worst = 0.0
    for n, p in eng.R.named_parameters():
        ref = float(g['step_gradnorm_' + n])
        if ref > 1e-3:      # biases in front of a train-mode BN have an (analytically) zero gradient
            worst = max(worst, abs(float(p.grad.norm()) - ref) / ref)
    print('cfg1 step: dS rows %.2e, dloggamma %.2e, worst R grad-norm rel diff %.2e' % (e_s, e_g, worst))
    assert e_s < 5e-3 and e_g < 5e-3 and worst < 5e-3
    assert rel_err(eng.S.LOGGAMMA.detach(), g['step_post_loggamma']) < 1e-5
    upd = float((eng.S.SUPPORT_SETS.detach().cpu() - c['sd']['SUPPORT_SETS']).abs().mean())
    assert abs(upd - float(g['step_post_S_absmean_update'])) < 0.02 * float(g['step_post_S_absmean_update'])
