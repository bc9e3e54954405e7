"""GPU: the full training step (warp -> G -> R -> loss -> backward -> Adam x2) vs the oracle's replay of
lib/trainer.py:190-254 on identical (z, idx, magnitudes) and identical initial weights."""
import types

import pytest
import torch

from oracle import wgs_oracle as O
from tests import golden_inputs as GI
from tests.util import rel_err
from warpedganspace_amd.gan_load import StyleGAN2Wrapper
from warpedganspace_amd.reconstructor import Reconstructor
from warpedganspace_amd.stylegan2 import Generator
from warpedganspace_amd.support_sets import SupportSets
from warpedganspace_amd.trainer import TrainStep

pytestmark = pytest.mark.gpu


def make(dev, size, K, N, B, w_space=False, precision='fp32', r_precision='auto'):
    torch.manual_seed(0)
    G = Generator(size, 512, 8)
    sd_g = GI.fill_state_dict(G.state_dict(), 900 + size)
    # keep the (random) mapping network well conditioned: scale its weights up so that w is O(1)
    for k in sd_g:
        if k.startswith('style.') and k.endswith('weight'):
            sd_g[k] = sd_g[k] * 100.0
    G.load_state_dict(sd_g)
    c = GI.support_sets_case(K, N, 512, B, 31, learn_gammas=True)
    S = SupportSets(K, N, 512, learn_gammas=True, gamma=c['gamma'])
    S.load_state_dict(c['sd'])
    R = Reconstructor('ResNet', K)
    sd_r = {k: v.detach().clone().contiguous() for k, v in R.state_dict().items()}
    params = types.SimpleNamespace(reconstructor_lr=1e-4, support_set_lr=1e-4, min_shift_magnitude=0.25,
                                   max_shift_magnitude=0.45, lambda_cls=1.0, lambda_reg=0.25, z_truncation=None,
                                   shift_in_w_space=w_space)
    ref = O.ReferenceStep(sd_g, c['sd'], sd_r, size, learn_gammas=True, gamma=c['gamma'], shift_in_w_space=w_space,
                          g_requires_grad=False)
    wrap = StyleGAN2Wrapper(G, w_space).to(dev).eval()
    eng = TrainStep(wrap, S.to(dev).train(), R.to(dev).train(), params, B, dev, seed=1, precision=precision, r_precision=r_precision)
    return eng, ref, c


@pytest.mark.parametrize('w_space', [False, True])
def test_two_steps_vs_reference_replay(dev, w_space):
    size, K, N, B = 32, 16, 4, 4
    eng, ref, c = make(dev, size, K, N, B, w_space)
    g = torch.Generator().manual_seed(7)
    first_idx, after_first = [], None
    for it in range(2):
        z = torch.randn(B, 512, generator=g)
        idx = torch.randint(0, K, (B,), generator=g)
        mag = (torch.rand(B, generator=g) * 0.2 + 0.25) * torch.where(torch.rand(B, generator=g) > 0.3, 1.0, -1.0)
        o = ref.step(z, idx, mag)
        stats = eng.step(z.to(dev), idx.to(dev), mag.to(dev)).tolist()
        # step 1 starts from identical weights: tight.  Step 2 starts after one Adam update, whose first
        # step is lr*sign(g) — entries with numerically-zero gradient may move either way in two fp32
        # evaluations, so the second forward can differ at the 1e-3 level (same for any two backends).
        tol = 1e-4 if it == 0 else 5e-2
        assert abs(stats[0] - o['ce']) < tol * max(1.0, abs(o['ce']))
        assert abs(stats[1] - o['l1']) < tol * max(1.0, abs(o['l1']))
        assert abs(stats[2] - o['loss']) < tol * max(1.0, abs(o['loss']))
        if it == 0:
            assert abs(stats[3] - o['acc']) < 1e-6
            assert torch.equal(eng.argmax.cpu(), o['argmax'])                   # path-index argmax bit-exact
        # gradients (read back from the flat bucket before they are reused) — well-posed comparison
        gb = eng.bucket.gview
        if it == 1:
            # dense Adam: a row selected in step 1 but not in step 2 still moves in step 2 (momentum)
            only_first = [k for k in first_idx if k not in set(idx.tolist())]
            if only_first:
                now = eng.S.state_dict()['SUPPORT_SETS'].cpu()[only_first]
                assert float((now - after_first[only_first]).abs().max()) > 1e-5
            break   # beyond this point the two fp32 trajectories have legitimately decorrelated
        gtol = 2e-3
        assert rel_err(gb[id(eng.S.SUPPORT_SETS)], ref.s['SUPPORT_SETS'].grad) < gtol, it
        assert rel_err(gb[id(eng.S.LOGGAMMA)], ref.s['LOGGAMMA'].grad) < gtol, it
        worst = 0.0
        for name, prm in eng.R.named_parameters():
            if name.startswith('features_extractor.fc'):
                continue
            gref = ref.r[name].grad
            gm = prm.grad
            worst = max(worst, rel_err(gm, gref))
        assert worst < gtol, (it, worst)
        # post-step parameters.  Adam's first steps move every weight by ~lr*sign(g): entries whose gradient
        # is numerically zero may legitimately go either way, so compare in the mean and by sign agreement.
        sd_s = eng.S.state_dict()
        rows = torch.unique(idx)
        upd_ref = ref.s['SUPPORT_SETS'].detach()[rows] - c['sd']['SUPPORT_SETS'][rows]
        upd = sd_s['SUPPORT_SETS'].cpu()[rows] - c['sd']['SUPPORT_SETS'][rows]
        assert float((torch.sign(upd) == torch.sign(upd_ref)).float().mean()) > (0.995 if it == 0 else 0.97)
        assert float((upd - upd_ref).abs().mean()) < 0.05 * 1e-4 * (it + 1)
        untouched = [k for k in range(K) if k not in set(idx.tolist()) and (it == 0)]
        if untouched:   # dense Adam: rows never selected so far have zero gradient and zero moments -> unchanged
            assert torch.equal(sd_s['SUPPORT_SETS'].cpu()[untouched], c['sd']['SUPPORT_SETS'][untouched])
        first_idx, after_first = idx.tolist(), sd_s['SUPPORT_SETS'].cpu().clone()
        sd_r = eng.R.state_dict()
        for k, v in ref.r.items():
            if k.startswith('features_extractor.fc') or k.endswith('num_batches_tracked'):
                continue
            d = (sd_r[k].cpu() - v.detach()).abs()
            assert float(d.mean()) < 0.05 * 1e-4 * (it + 1) + 1e-6 * float(v.detach().abs().mean()) + 1e-7, (it, k, float(d.mean()))
    st = eng.pop_stats()
    assert set(st) == {'accuracy', 'classification_loss', 'regression_loss', 'total_loss'}


def test_sampler_distribution_matches_reference_quirk(dev):
    """lib/trainer.py:212-221: weights = arange(2B) without replacement => ~71 % positive magnitudes and
    index 0 (the first negative one) is never drawn; |mag| in [min, max]."""
    eng, _, _ = make(dev, 32, 16, 4, 64)
    pos = tot = 0
    for _ in range(50):
        z, idx, mag = eng.sample()
        assert z.shape == (64, 512) and idx.min() >= 0 and idx.max() < 16
        a = mag.abs()
        assert float(a.min()) >= 0.25 - 1e-6 and float(a.max()) <= 0.45 + 1e-6
        pos += int((mag > 0).sum())
        tot += 64
    assert 0.64 < pos / tot < 0.78


@pytest.mark.parametrize('tail', [False, True])
def test_prefetched_unshifted_pass_same_trajectory(dev, tail):
    """The next step's un-shifted pass G(z) is drawn and generated one step ahead on a third stream (trainer.py): the
    sampler's draws — hence every batch — are bit-identical to the plain schedule's, the first two steps' statistics agree
    to rounding, and the trajectory stays inside the run-to-run envelope of the plain schedule itself.  tail: the pass in THREE
    stages, its last layers enqueued from inside the generator's backward (here: pauses above 8^2 and 16^2, hook at <= 8^2)."""
    size, K, N, B = 32, 16, 4, 4
    runs = []
    for prefetch in (False, True):
        eng, _, _ = make(dev, size, K, N, B)
        eng.prefetch = prefetch
        eng.tail_prefetch = tail
        eng.split_pause_res, eng.tail_pause_res, eng.tail_hook_res = 8, 16, 8
        drawn, plain_sample = [], eng.sample

        def sample(_s=plain_sample, _d=drawn):
            b = _s()
            _d.append(tuple(t.clone() for t in b))
            return b
        eng.sample = sample
        traj, ahead = [], 0
        for it in range(5):
            ahead += eng._pre is not None
            traj.append(eng.step().tolist())
        torch.cuda.synchronize()
        runs.append((traj, drawn, ahead, eng.bucket.flat.detach().clone()))
        assert (eng._pre is not None) == prefetch
    (t0, d0, a0, p0), (t1, d1, a1, p1) = runs
    assert a0 == 0 and a1 == 3        # steps 3..5 consumed a batch generated one step ahead (the first step runs single-stream)
    assert len(d0) == 5 and len(d1) == 6          # one batch stays unused when the prefetching run stops
    for b0, b1 in zip(d0, d1):
        assert all(torch.equal(x, y) for x, y in zip(b0, b1))
    for x, y in zip(t0[0], t1[0]):
        assert abs(x - y) <= 1e-6 * max(1.0, abs(x)), (t0[0], t1[0])
    for x, y in zip(t0[1][:3], t1[1][:3]):
        assert abs(x - y) <= 1e-4 * max(1.0, abs(x)), (t0[1], t1[1])
    # from the third step on two runs of the SAME schedule already land on one of two branches (3.03137 / 3.03095 ...: an
    # Adam sign step on a numerically-zero gradient, measured with tools-level repeats of the plain schedule), so only the
    # envelope of the step-2 test above applies
    for a, b in zip(t0, t1):
        for x, y in zip(a[:3], b[:3]):
            assert abs(x - y) <= 5e-2 * max(1.0, abs(x)), (t0, t1)
    assert rel_err(p1, p0) < 1e-3


def test_runtime_precision_check_and_engines_with_different_modes_coexist(dev):
    """TrainStep.check_precision measures the engine's 16-bit mode against the exact-fp32 kernels on ITS generator's weights
    (the guard behind train.py --check-precision); set_precision switches one engine only: two engines on the same generator,
    one fp32 and one fp16, keep producing their own arithmetic (no process-wide state)."""
    from warpedganspace_amd import conv as C
    size, K, N, B = 32, 16, 4, 4
    e32, _, _ = make(dev, size, K, N, B, precision='fp32')
    e16 = TrainStep(e32.G, e32.S, e32.R, e32.p, B, dev, seed=1, precision='f16x2')
    assert e32.precision == 0 and e16.precision == 3 and tuple(e32.r_arith) == (0, 0, 0) and tuple(e16.r_arith) == (1, 1, 1)
    assert e32.check_precision() is None
    r = e16.check_precision()
    assert r['precision'] == 'f16x2' and r['n'] == B and 0 < r['batch'] < 1e-3 and r['ok'] and r['per_image_max'] >= r['per_image_median']
    z = torch.randn(B, 512, device=dev)
    with torch.no_grad():
        i32, i16 = e32.G(z, precision=e32.precision), e16.G(z, precision=e16.precision)
        again32 = e32.G(z, precision=e32.precision)
    assert torch.equal(i32, again32) and not torch.equal(i32, i16) and rel_err(i16, i32) < 1e-3
    e16.step(); e32.step(); e16.step()                     # interleaved steps of the two engines
    torch.cuda.synchronize()
    # a mode that misses the gate is replaced: the Trainer's policy, exercised through its hook with a gate of ~0
    r0 = e16.check_precision(gate=1e-12)
    assert not r0['ok']
    e16.set_precision(C.AUTO_FALLBACK)
    assert e16.precision == 1 and e16._pre is None and e16._cold
    assert e16.check_precision()['batch'] < 1e-4
    e16.step()
    torch.cuda.synchronize()
    assert all(v == v for v in e16.stats.tolist())
