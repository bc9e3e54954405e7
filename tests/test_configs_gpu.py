"""GPU: the BASELINE.json configurations at (or near) their stated sizes, beyond the small-size parity tests:

  * cfg3 image size: one training step on StyleGAN2-256 (256x256 images through ResNet-18) against the oracle's replay of
    lib/trainer.py:190-254, in exact fp32 and in the architecture's default arithmetic (f16);
  * cfg2: TrainStep with a ProgGAN generator (K=64, N=16) vs the replay;
  * cfg4: TrainStep with the BigGAN-128 reference architecture vs the replay; BigGAN-256 (generator_arch 256, class-conditional,
    B=16) forward vs the oracle and batch-16 forward / input gradient across arithmetic modes;
  * cfg5 as specified: StyleGAN2-1024, K=200, N=64, B=8, fp16 MFMA path — a full-size step, checked against the same step
    in exact fp32 on the GPU (the exact kernels are pinned to the oracle at the sizes the CPU can replay).
"""
import types

import pytest
import torch

from oracle import wgs_oracle as O
from tests import golden_inputs as GI
from tests.util import rel_err
from warpedganspace_amd import conv as C
from warpedganspace_amd.reconstructor import Reconstructor
from warpedganspace_amd.support_sets import SupportSets
from warpedganspace_amd.trainer import TrainStep

pytestmark = pytest.mark.gpu


def _params(w_space=False):
    return types.SimpleNamespace(reconstructor_lr=1e-4, support_set_lr=1e-4, min_shift_magnitude=0.25, max_shift_magnitude=0.45,
                                 lambda_cls=1.0, lambda_reg=0.25, z_truncation=None, shift_in_w_space=w_space)


def _samples(B, d, K, seed):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(B, d, generator=g)
    idx = torch.randint(0, K, (B,), generator=g)
    mag = (torch.rand(B, generator=g) * 0.2 + 0.25) * torch.where(torch.rand(B, generator=g) > 0.3, 1.0, -1.0)
    return z, idx, mag


def _check_step(eng, ref, z, idx, mag, dev, loss_tol, grad_tol):
    o = ref.step(z, idx, mag)
    st = eng.step(z.to(dev), idx.to(dev), mag.to(dev)).tolist()
    gb = eng.bucket.gview
    e_s = rel_err(gb[id(eng.S.SUPPORT_SETS)], ref.s['SUPPORT_SETS'].grad)
    a, b = gb[id(eng.S.SUPPORT_SETS)].double().cpu().reshape(-1), ref.s['SUPPORT_SETS'].grad.double().reshape(-1)
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    print('   loss %.6f (oracle %.6f)  ce %.6f/%.6f  l1 %.6f/%.6f  dS max-norm err %.2e cosine %.6f' % (
        st[2], o['loss'], st[0], o['ce'], st[1], o['l1'], e_s, cos))
    assert abs(st[2] - o['loss']) < loss_tol * max(1.0, abs(o['loss']))
    assert torch.equal(eng.argmax.cpu(), o['argmax'])                     # path-index argmax bit-exact
    # free-running gradients through a batch of 2..4 with train-mode BatchNorm and ~1e6 activation gates: individual entries
    # move by a few per cent between ANY two fp32 evaluations (max-norm), the direction does not
    if grad_tol is not None:
        assert e_s < grad_tol, e_s
    assert cos > 0.999
    return st, o


@pytest.fixture(scope='module')
def sg2_256_case():
    """StyleGAN2-256, K=8, N=2, B=2, ResNet-18 on 256x256 images: weights, samples and the oracle's replay (CPU, ~1 min)."""
    from warpedganspace_amd.stylegan2 import Generator
    size, K, N, B = 256, 8, 2, 2
    torch.manual_seed(0)
    G = Generator(size, 512, 8)
    sd_g = GI.fill_state_dict(G.state_dict(), 900 + size)
    for k in sd_g:
        if k.startswith('style.') and k.endswith('weight'):
            sd_g[k] = sd_g[k] * 100.0
    c = GI.support_sets_case(K, N, 512, B, 31, learn_gammas=True)
    R = Reconstructor('ResNet', K)
    sd_r = {k: v.detach().clone().contiguous() for k, v in R.state_dict().items()}
    z, idx, mag = _samples(B, 512, K, 11)
    ref = O.ReferenceStep(sd_g, c['sd'], sd_r, size, learn_gammas=True, gamma=c['gamma'], g_requires_grad=False)
    o = ref.step(z, idx, mag)
    grads = {'S': ref.s['SUPPORT_SETS'].grad.clone(), 'lg': ref.s['LOGGAMMA'].grad.clone(),
             'R': {k: v.grad.clone() for k, v in ref.r.items() if v.is_floating_point() and v.requires_grad and v.grad is not None}}
    return dict(size=size, K=K, N=N, B=B, sd_g=sd_g, c=c, sd_r=sd_r, z=z, idx=idx, mag=mag, o=o, grads=grads)


@pytest.mark.parametrize('mode', ['fp32', 'fp32w', 'bf16x3', 'auto'])
def test_step_at_256x256_images_vs_oracle_replay(dev, sg2_256_case, mode):
    from warpedganspace_amd.gan_load import StyleGAN2Wrapper
    from warpedganspace_amd.stylegan2 import Generator
    t = sg2_256_case
    G = Generator(t['size'], 512, 8)
    G.load_state_dict(t['sd_g'])
    S = SupportSets(t['K'], t['N'], 512, learn_gammas=True, gamma=t['c']['gamma'])
    S.load_state_dict(t['c']['sd'])
    R = Reconstructor('ResNet', t['K'])
    R.load_state_dict(t['sd_r'])
    eng = TrainStep(StyleGAN2Wrapper(G, False).to(dev).eval(), S.to(dev).train(), R.to(dev).train(), _params(), t['B'], dev, seed=1,
                    precision=mode)
    st = eng.step(t['z'].to(dev), t['idx'].to(dev), t['mag'].to(dev)).tolist()
    o, gr = t['o'], t['grads']
    gb = eng.bucket.gview
    e_s = rel_err(gb[id(eng.S.SUPPORT_SETS)], gr['S'])
    worst = max(rel_err(prm.grad, gr['R'][n]) for n, prm in eng.R.named_parameters() if n in gr['R'] and not n.startswith('features_extractor.fc'))
    name = C.precision_name(eng.precision)
    print('StyleGAN2-256 step, %s: loss %.6f (oracle %.6f), dS err %.2e, worst dR err %.2e' % (name, st[2], o['loss'], e_s, worst))
    tight = mode in ('fp32', 'fp32w', 'bf16x3')
    assert abs(st[2] - o['loss']) < (1e-4 if tight else 3e-4) * max(1.0, abs(o['loss']))
    assert abs(st[0] - o['ce']) < (1e-4 if tight else 3e-4) * max(1.0, abs(o['ce']))
    assert torch.equal(eng.argmax.cpu(), o['argmax'])
    a, b = gb[id(eng.S.SUPPORT_SETS)].double().cpu().reshape(-1), gr['S'].double().reshape(-1)
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    print('   dS cosine %.6f' % cos)
    assert cos > (0.999 if tight else 0.98)     # batch of 2 through train-mode BatchNorm: entries move by per cents, the direction holds
    # max-norm errors of the gradients themselves (asserted, not only printed).  The floor is set by the evaluation, not
    # the arithmetic: with a batch of 2 through train-mode BatchNorm a handful of ReLU gates / max-pool winners that round to the
    # other side move single entries by per cents in ANY two evaluations (exact fp32 measures dS 1.8e-2 / worst dR 7.7e-2 against the
    # fp64 oracle; the Winograd form the same 1.8e-2 / 7.7e-2; split-bf16 3.7e-2 / 1.4e-1; the mixed fp16 policy 1.7e-1 / 2.5e-1)
    lim_s, lim_r = {'fp32': (4e-2, 1.5e-1), 'fp32w': (4e-2, 1.5e-1), 'bf16x3': (8e-2, 2.7e-1)}.get(mode, (3.5e-1, 5e-1))
    assert e_s < lim_s and worst < lim_r, (mode, e_s, worst)


def test_trainstep_proggan_k64_n16_vs_replay(dev):
    """cfg2's support-set shape (K=64, N=16) on a ProgGAN truncated to 10 blocks (64x64, the 512/256-channel layers)."""
    from tests.test_oracle_golden import _proggan_sd
    from warpedganspace_amd.proggan import Generator, ProgGANWrapper
    K, N, B, nb = 64, 16, 4, 10
    sd_g = _proggan_sd(610, nb)
    G = Generator(nb)
    G.load_state_dict(sd_g)
    c = GI.support_sets_case(K, N, 512, B, 611, learn_gammas=True)
    S = SupportSets(K, N, 512, learn_gammas=True, gamma=c['gamma'])
    S.load_state_dict(c['sd'])
    torch.manual_seed(5)
    R = Reconstructor('ResNet', K)
    sd_r = {k: v.detach().clone().contiguous() for k, v in R.state_dict().items()}
    ref = O.ReferenceStep(sd_g, c['sd'], sd_r, 64, learn_gammas=True, gamma=c['gamma'], generator='ProgGAN', gen_kwargs=dict(num_blocks=nb),
                          g_requires_grad=False)
    eng = TrainStep(ProgGANWrapper(G).to(dev).eval(), S.to(dev).train(), R.to(dev).train(), _params(), B, dev, seed=2)
    z, idx, mag = _samples(B, 512, K, 612)
    print('ProgGAN (10 blocks) K=64 N=16 step:')
    _check_step(eng, ref, z, idx, mag, dev, 1e-4, 5e-2)


def test_trainstep_biggan128_vs_replay(dev):
    """cfg4's generator family: the reference's BigGAN-128 architecture (class 239) inside the training step."""
    from tests.test_oracle_golden import _biggan
    K, N, B = 16, 4, 4
    W = _biggan()
    sd_g = {k: v.detach().clone() for k, v in W.G.state_dict().items()}
    c = GI.support_sets_case(K, N, 120, B, 621, learn_gammas=True)
    S = SupportSets(K, N, 120, learn_gammas=True, gamma=c['gamma'])
    S.load_state_dict(c['sd'])
    torch.manual_seed(6)
    R = Reconstructor('ResNet', K)
    sd_r = {k: v.detach().clone().contiguous() for k, v in R.state_dict().items()}
    ref = O.ReferenceStep(sd_g, c['sd'], sd_r, 128, learn_gammas=True, gamma=c['gamma'], generator='BigGAN',
                          gen_kwargs=dict(class_ids=torch.full((B,), 239)), g_requires_grad=False)
    eng = TrainStep(W.to(dev).eval(), S.to(dev).train(), R.to(dev).train(), _params(), B, dev, seed=3)
    z, idx, mag = _samples(B, 120, K, 622)
    print('BigGAN-128 step:')
    _check_step(eng, ref, z, idx, mag, dev, 1e-4, 5e-2)


def test_biggan256_class_conditional_batch16(dev):
    """cfg4 as stated (256x256, class-conditional, B=16): generator_arch[256] (6 blocks, attention at 64x64, z chunks of 17)
    forward vs the oracle on 2 samples, then batch 16: fp32 vs bf16x3 forward and input gradient, batch consistency."""
    from warpedganspace_amd.biggan import BigGANWrapper, Generator
    torch.manual_seed(9)
    G = Generator(G_ch=96, dim_z=120, shared_dim=128, hier=True, G_attn='64', BN_eps=1e-5, SN_eps=1e-6, resolution=256, n_classes=1000)
    G.load_state_dict(GI.fill_state_dict(G.state_dict(), 640, fan_in=True, per_key=True))
    assert G.dim_z == 119 and len(G.blocks) == 6
    sd = {k: v.detach().clone() for k, v in G.state_dict().items()}
    G = G.to(dev).eval()
    z = GI.rt(641, 16, 119)
    cls = torch.tensor([239, 14] * 8)
    with torch.no_grad():
        ref2 = O.biggan_generate(sd, z[:2], cls[:2], resolution=256)
    outs, grads = {}, {}
    probe = GI.rt(643, 16, 3, 256, 256).to(dev)
    for mode in ('fp32', 'bf16x3'):
        sh = (GI.rt(642, 16, 119) * 0.1).to(dev).requires_grad_(True)
        img = G(z.to(dev) + sh, G.shared(cls.to(dev)), precision=mode)
        (img * probe).sum().backward()
        outs[mode], grads[mode] = img.detach(), sh.grad.detach()
    with torch.no_grad():
        img2 = G(z[:2].to(dev), G.shared(cls[:2].to(dev)))
    e_or = rel_err(img2, ref2)
    e_mode, e_grad = rel_err(outs['bf16x3'], outs['fp32']), rel_err(grads['bf16x3'], grads['fp32'])
    print('BigGAN-256: forward vs oracle %.2e; batch 16: bf16x3 vs fp32 image %.2e, input gradient %.2e' % (e_or, e_mode, e_grad))
    assert outs['fp32'].shape == (16, 3, 256, 256)
    assert e_or < 1e-4 and e_mode < 1e-4
    assert e_grad < 5e-2              # free-running gradients: ReLU gate flips between two fp32-class evaluations
    W = BigGANWrapper(G, (239,))
    assert W.dim_z == 119 and W(torch.randn(2, 119, device=dev)).shape == (2, 3, 256, 256)


def test_cfg5_full_size_step_fp16_path(dev):
    """cfg5 as specified: StyleGAN2-1024, K=200, N=64, batch 8, the fp16 MFMA path = this architecture's 'mixed' policy (what `auto`
    resolves to and what bench.py's cfg5 line runs; its image error against the fp64 oracle is gated in
    test_precision_schemes_gpu.py) — one full-size training step against the same step in exact fp32 (identical samples and
    initial weights): loss within 1e-3, argmax bit-exact, finite gradients, and the support-set gradient pointing the same way."""
    from warpedganspace_amd.gan_load import build_stylegan2
    K, N, B = 200, 64, 8
    res = {}
    for mode in ('fp32', 'auto'):
        torch.manual_seed(0)
        G = build_stylegan2(None, resolution=1024)
        sd = G.G.state_dict()
        for k in sd:      # keep the random mapping network from collapsing every z onto one w
            if k.startswith('style.') and k.endswith('weight'):
                sd[k] = sd[k] * 100.0
        G.G.load_state_dict(sd)
        S = SupportSets(K, N, 512, learn_gammas=True, gamma=1.0 / 512)
        R = Reconstructor('ResNet', K)
        eng = TrainStep(G.to(dev).eval(), S.to(dev).train(), R.to(dev).train(), _params(), B, dev, seed=4, precision=mode)
        z, idx, mag = _samples(B, 512, K, 650)
        st = eng.step(z.to(dev), idx.to(dev), mag.to(dev)).tolist()
        res[mode] = (st, eng.argmax.cpu().clone(), eng.bucket.gview[id(eng.S.SUPPORT_SETS)].double().cpu().reshape(-1).clone())
        del eng, G, S, R
        torch.cuda.empty_cache()
    assert C.resolve('auto', 'stylegan2', 1024) == C.MIXED_STRICT       # the table the engine calibrated on this generator
    (s0, a0, g0), (s1, a1, g1) = res['fp32'], res['auto']
    cos = float((g0 * g1).sum() / (g0.norm() * g1.norm()))
    print('cfg5 step (1024^2, K=200, N=64, B=8): loss fp32 %.6f mixed %.6f, dS cosine %.5f' % (s0[2], s1[2], cos))
    assert all(v == v for v in s1) and torch.isfinite(g1).all()
    assert abs(s1[2] - s0[2]) < 1e-3 * max(1.0, abs(s0[2]))
    assert torch.equal(a0, a1)
    assert cos > 0.9
