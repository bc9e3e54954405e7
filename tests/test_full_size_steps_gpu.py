"""GPU: the BASELINE.json configurations at their FULL sizes (the sizes bench.py times), where no CPU oracle can replay the step:
size-independent properties of one training step from identical weights and samples —
  * the product's default arithmetic (`auto`) against the exact-fp32 step (itself pinned to the oracle at the sizes the CPU can
    replay: tests/test_configs_gpu.py, test_train_step_gpu.py): loss / CE / L1 within 1e-3, path-index argmax bit-exact, the
    support-set gradient pointing the same way, everything finite;
  * a second identical fp32 step reproduces the first one's loss and argmax (no stale state, no cross-step races);
  * the Adam update moved the parameters, and only by ~lr.
cfg2 ProgGAN-1024 (K=64, N=16, B=32), cfg3 StyleGAN2-256 (K=128, N=32, B=32), cfg4 BigGAN-256 class-conditional (K=128, N=32, B=16).
(cfg5 at full size: tests/test_configs_gpu.py::test_cfg5_full_size_step_fp16_path; cfg1 is the CPU-sized golden test.)"""
import types

import pytest
import torch

from warpedganspace_amd import conv as C
from warpedganspace_amd.reconstructor import Reconstructor
from warpedganspace_amd.support_sets import SupportSets
from warpedganspace_amd.trainer import TrainStep

pytestmark = pytest.mark.gpu


def _params():
    return types.SimpleNamespace(reconstructor_lr=1e-4, support_set_lr=1e-4, min_shift_magnitude=0.25, max_shift_magnitude=0.45,
                                 lambda_cls=1.0, lambda_reg=0.25, z_truncation=None, shift_in_w_space=False)


def _generator(cfg):
    torch.manual_seed(0)
    if cfg == 'cfg3':
        from warpedganspace_amd.gan_load import build_stylegan2
        G = build_stylegan2(None, resolution=256)
        sd = G.G.state_dict()
        for k in sd:      # keep the random mapping network from collapsing every z onto one w
            if k.startswith('style.') and k.endswith('weight'):
                sd[k] = sd[k] * 100.0
        G.G.load_state_dict(sd)
        return G, 128, 32, 32
    if cfg == 'cfg2':
        from warpedganspace_amd.proggan import build_proggan
        return build_proggan(None, num_blocks=18), 64, 16, 32
    from warpedganspace_amd.biggan import BigGANWrapper, Generator
    G = Generator(G_ch=96, dim_z=120, shared_dim=128, hier=True, G_attn='64', BN_eps=1e-5, SN_eps=1e-6, resolution=256, n_classes=1000)
    return BigGANWrapper(G, (239,)), 128, 32, 16


def _fp32_class_agreement(cfg, a, w):
    cw = float((a['gS'] * w['gS']).sum() / (a['gS'].norm() * w['gS'].norm()))
    print('%s full-size step, fp32w vs fp32: loss %.6f vs %.6f, CE %.6f vs %.6f, L1 %.6f vs %.6f, dS cosine %.6f' % (
        cfg, w['st'][2], a['st'][2], w['st'][0], a['st'][0], w['st'][1], a['st'][1], cw))
    for i in range(3):
        assert abs(w['st'][i] - a['st'][i]) < 1e-5 * max(1.0, abs(a['st'][i])), (cfg, i, a['st'], w['st'])
    assert torch.equal(a['argmax'], w['argmax'])
    assert cw > 0.9999


def _one_step(dev, G, K, N, B, mode, seed=77):
    torch.manual_seed(1)
    S = SupportSets(K, N, G.dim_z, learn_alphas=False, learn_gammas=True, gamma=1.0 / G.dim_z)
    R = Reconstructor('ResNet', K)
    eng = TrainStep(G.to(dev).eval(), S.to(dev).train(), R.to(dev).train(), _params(), B, dev, seed=4, precision=mode)
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(B, G.dim_z, generator=g)
    idx = torch.randint(0, K, (B,), generator=g)
    mag = (torch.rand(B, generator=g) * 0.2 + 0.25) * torch.where(torch.rand(B, generator=g) > 0.3, 1.0, -1.0)
    st = eng.step(z.to(dev), idx.to(dev), mag.to(dev)).tolist()
    torch.cuda.synchronize()
    assert all(v == v and abs(v) < 1e6 for v in st) and torch.isfinite(eng.bucket.grad).all() and torch.isfinite(eng.bucket.flat).all()
    out = dict(st=st, argmax=eng.argmax.cpu().clone(), gS=eng.bucket.gview[id(eng.S.SUPPORT_SETS)].double().cpu().reshape(-1).clone(),
               prec=C.precision_name(eng.precision))
    del eng, S, R
    torch.cuda.empty_cache()
    return out


def test_cfg5_full_size_step_winograd_form_vs_direct_fp32(dev):
    """cfg5 at full size (StyleGAN2-1024, K=200, N=64, B=8): the Winograd form of the 3x3 stride-1 convs (the 64-channel 512^2 layers
    take the 64-channel workgroup shape, the 32-channel 1024^2 layers stay direct) against direct-form exact fp32."""
    from warpedganspace_amd.gan_load import build_stylegan2
    res = {}
    for mode in ('fp32', 'fp32w'):
        torch.manual_seed(0)
        G = build_stylegan2(None, resolution=1024)
        sd = G.G.state_dict()
        for k in sd:
            if k.startswith('style.') and k.endswith('weight'):
                sd[k] = sd[k] * 100.0
        G.G.load_state_dict(sd)
        res[mode] = _one_step(dev, G, 200, 64, 8, mode)
        del G
        torch.cuda.empty_cache()
    assert res['fp32w']['prec'] == 'fp32w'
    _fp32_class_agreement('cfg5', res['fp32'], res['fp32w'])


def test_cfg4_as_benched_full_size_step(dev):
    """cfg4 exactly as bench.py times it: BigGAN-128 (the reference's architecture), K=128, N=32, B=16 — the product's default
    arithmetic for this generator (bf16x3) against exact fp32 from identical weights and samples (the oracle replays this architecture
    at K=16, N=4, B=4: tests/test_configs_gpu.py)."""
    from warpedganspace_amd.biggan import BigGANWrapper, Generator
    res = {}
    for mode in ('fp32', 'auto'):
        torch.manual_seed(0)
        G = BigGANWrapper(Generator(G_ch=96, dim_z=120, shared_dim=128, hier=True, G_attn='64', BN_eps=1e-5, SN_eps=1e-6, resolution=128,
                                    n_classes=1000), (239,))
        res[mode] = _one_step(dev, G, 128, 32, 16, mode)
        del G
        torch.cuda.empty_cache()
    a, b = res['fp32'], res['auto']
    cos = float((a['gS'] * b['gS']).sum() / (a['gS'].norm() * b['gS'].norm()))
    print('cfg4 BigGAN-128 K=128 N=32 B=16 step: loss fp32 %.6f | %s %.6f ; dS cosine %.5f' % (a['st'][2], b['prec'], b['st'][2], cos))
    for i in range(3):
        assert abs(b['st'][i] - a['st'][i]) < 1e-3 * max(1.0, abs(a['st'][i])), (i, a['st'], b['st'])
    assert torch.equal(a['argmax'], b['argmax'])
    assert cos > 0.98


@pytest.mark.parametrize('cfg', ['cfg3', 'cfg2', 'cfg4'])
def test_full_size_step_default_arithmetic_vs_exact_fp32(dev, cfg):
    res = {}
    # cfg3: the headline bench mode as well (fp32w: at B = 32 every 3x3 stride-1 layer shape takes wino_f32_kernel<1, 4, true, 8>)
    for mode in ('fp32', 'fp32-again', 'auto') + (('fp32w',) if cfg == 'cfg3' else ()):
        G, K, N, B = _generator(cfg)
        torch.manual_seed(1)
        S = SupportSets(K, N, G.dim_z, learn_alphas=False, learn_gammas=True, gamma=1.0 / G.dim_z)
        R = Reconstructor('ResNet', K)
        eng = TrainStep(G.to(dev).eval(), S.to(dev).train(), R.to(dev).train(), _params(), B, dev, seed=4, precision=mode.split('-')[0])
        g = torch.Generator().manual_seed(77)
        z = torch.randn(B, G.dim_z, generator=g)
        idx = torch.randint(0, K, (B,), generator=g)
        mag = (torch.rand(B, generator=g) * 0.2 + 0.25) * torch.where(torch.rand(B, generator=g) > 0.3, 1.0, -1.0)
        before = eng.bucket.flat.detach().clone()
        st = eng.step(z.to(dev), idx.to(dev), mag.to(dev)).tolist()
        torch.cuda.synchronize()
        moved = (eng.bucket.flat.detach() - before).abs()
        res[mode] = dict(st=st, argmax=eng.argmax.cpu().clone(), gS=eng.bucket.gview[id(eng.S.SUPPORT_SETS)].double().cpu().reshape(-1).clone(),
                         prec=C.precision_name(eng.precision), moved=(float(moved.max()), float((moved > 0).float().mean())))
        assert all(v == v and abs(v) < 1e6 for v in st) and torch.isfinite(eng.bucket.grad).all() and torch.isfinite(eng.bucket.flat).all()
        del eng, G, S, R
        torch.cuda.empty_cache()
    a, a2, b = res['fp32'], res['fp32-again'], res['auto']
    cos = float((a['gS'] * b['gS']).sum() / (a['gS'].norm() * b['gS'].norm()))
    print('%s full-size step: loss fp32 %.6f | again %.6f | %s %.6f ; dS cosine %.5f ; largest Adam move %.2e (%.0f %% of the parameters moved)' % (
        cfg, a['st'][2], a2['st'][2], b['prec'], b['st'][2], cos, a['moved'][0], 100 * a['moved'][1]))
    # the same step twice: identical up to the fp32 atomics of the style / weight-gradient reductions
    assert abs(a2['st'][2] - a['st'][2]) < 1e-5 * max(1.0, abs(a['st'][2])) and torch.equal(a['argmax'], a2['argmax'])
    for i in range(3):      # CE, L1, total
        assert abs(b['st'][i] - a['st'][i]) < 1e-3 * max(1.0, abs(a['st'][i])), (cfg, i, a['st'], b['st'])
    assert torch.equal(a['argmax'], b['argmax'])
    assert cos > 0.98
    if 'fp32w' in res:       # fp32 throughout (16 instead of 36 multiplies per 2x2 outputs): fp32-class agreement with the direct form
        _fp32_class_agreement(cfg, a, res['fp32w'])
    # first Adam step = lr * sign(g) (bias-corrected): no parameter moves by more than ~lr, and most of them move
    assert 0 < a['moved'][0] <= 1.01e-4 and a['moved'][1] > 0.5
