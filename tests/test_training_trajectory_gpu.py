"""GPU: 400 training iterations end to end (sample -> warp -> G -> R -> loss -> backward -> Adam, the loop of lib/trainer.py:184-261)
on a small configuration: the losses go down, and the trajectory in a 16-bit generator arithmetic tracks the exact-fp32 one — a
property no single-step comparison shows (a wrong-signed or mis-scaled gradient anywhere in S, G's input gradient, R or Adam makes
the curves part within tens of iterations)."""
import types

import pytest
import torch

from warpedganspace_amd.gan_load import build_stylegan2
from warpedganspace_amd.reconstructor import Reconstructor
from warpedganspace_amd.support_sets import SupportSets
from warpedganspace_amd.trainer import TrainStep

pytestmark = pytest.mark.gpu


def _run(dev, mode, iters=400, size=32, K=8, N=4, B=16):
    torch.manual_seed(0)
    G = build_stylegan2(None, resolution=size)
    sd = G.G.state_dict()
    for k in sd:          # keep the random mapping network from collapsing every z onto one w
        if k.startswith('style.') and k.endswith('weight'):
            sd[k] = sd[k] * 100.0
    G.G.load_state_dict(sd)
    S = SupportSets(K, N, 512, learn_alphas=False, learn_gammas=True, gamma=1.0 / 512)
    R = Reconstructor('ResNet', K)
    # lr 5e-4 (the reference's recipes use 1e-4 for 100 000+ iterations): within 400 iterations the losses move by ten times their run-to-run scatter
    p = types.SimpleNamespace(reconstructor_lr=5e-4, support_set_lr=5e-4, min_shift_magnitude=0.25, max_shift_magnitude=0.45, lambda_cls=1.0,
                              lambda_reg=0.25, z_truncation=None, shift_in_w_space=False)
    s0 = S.SUPPORT_SETS.detach().clone()
    eng = TrainStep(G.to(dev).eval(), S.to(dev).train(), R.to(dev).train(), p, B, dev, seed=3, precision=mode)
    windows = []
    for it in range(iters):
        eng.step()
        if (it + 1) % 50 == 0:
            st = eng.pop_stats()
            windows.append((st['classification_loss'], st['regression_loss']))
    moved = float((eng.S.SUPPORT_SETS.detach().cpu() - s0).abs().max())
    return windows, moved


def test_losses_fall_and_16bit_trajectory_tracks_fp32(dev):
    w32, m32 = _run(dev, 'fp32')
    w16, m16 = _run(dev, 'f16x2')
    print('fp32 :', ' '.join('%.3f/%.3f' % w for w in w32), ' support sets moved %.3e' % m32)
    print('f16x2:', ' '.join('%.3f/%.3f' % w for w in w16), ' support sets moved %.3e' % m16)
    for w in (w32, w16):
        ce_first, reg_first = w[0][0], w[0][1]
        ce_last, reg_last = sum(x[0] for x in w[-4:]) / 4, sum(x[1] for x in w[-2:]) / 2
        assert all(v == v for x in w for v in x)
        # The run is not bit-reproducible (fp64 atomics in the BatchNorm statistics, then 400 chaotic steps): its 50-step means scatter by +-0.02.
        # At lr 1e-4 the classification loss moved by only -0.03 .. -0.06 in 400 iterations and the assertion had to be loosened to -0.012 (ADVICE
        # r5: barely above the noise).  At lr 5e-4 (round 6, four trajectories per arithmetic): classification loss 2.39-2.44 in the first window
        # -> 2.12-2.14 over the second half (ln 8 = 2.079 is chance: the drop is the classifier's calibration), regression loss 0.45-0.47 ->
        # 0.24-0.25 — below the 0.35 of the best sign-blind predictor of a magnitude in +-[0.25, 0.45]: R has learned which way the codes moved.
        assert ce_last < ce_first - 0.15, w
        assert reg_last < 0.7 * reg_first and reg_last < 0.30, w
    # the warping functions are being trained too (Adam moves every touched entry by ~lr per step)
    assert 5e-3 < m32 < 0.2 and 5e-3 < m16 < 0.2
    # same samples (same sampler seed), same initial weights: the 50-step means of the two arithmetics stay together
    for (c32, r32), (c16, r16) in zip(w32, w16):
        assert abs(c16 - c32) < 0.04 * c32 and abs(r16 - r32) < 0.12 * r32 + 0.01, (w32, w16)
