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


@pytest.mark.parametrize('cfg', ['cfg3', 'cfg2', 'cfg4'])
def test_full_size_step_default_arithmetic_vs_exact_fp32(dev, cfg):
    res = {}
    for mode in ('fp32', 'fp32-again', 'auto'):
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
    # first Adam step = lr * sign(g) (bias-corrected): no parameter moves by more than ~lr, and most of them move
    assert 0 < a['moved'][0] <= 1.01e-4 and a['moved'][1] > 0.5
