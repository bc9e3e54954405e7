"""CPU: the oracle restatements reproduce the golden vectors generated from the reference modules
(tools/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import wgs_oracle as O
from tests import golden_inputs as GI
from tests.util import rel_err

SS_CASES = {'tiny': (4, 2, 8, 3, 11), 'cfg1': (32, 8, 128, 4, 12), 'cfg3': (128, 32, 512, 4, 13),
            'cfg4': (16, 4, 120, 5, 14)}


@pytest.mark.parametrize('name', list(SS_CASES))
@pytest.mark.parametrize('lg', [True, False])
def test_support_sets_torch_oracle_vs_reference(golden, name, lg):
    g = golden('support_sets')
    K, N, d, B, seed = SS_CASES[name]
    c = GI.support_sets_case(K, N, d, B, seed, learn_gammas=lg)
    sd = {k: v.clone().requires_grad_(True) for k, v in c['sd'].items()}
    z = c['z'].clone().requires_grad_(True)
    y = O.support_sets_forward(sd, GI.one_hot(c['idx'], K), z, lg, c['gamma'])
    (y * c['gout']).sum().backward()
    tag = '%s_%s' % (name, 'lg' if lg else 'cg')
    assert rel_err(y, g[tag + '_out']) < 1e-6
    assert rel_err(z.grad, g[tag + '_dz']) < 1e-5
    assert rel_err(sd['ALPHAS'].grad, g[tag + '_dalphas']) < 1e-5
    if lg:
        assert rel_err(sd['LOGGAMMA'].grad, g[tag + '_dloggamma']) < 1e-4


@pytest.mark.parametrize('name', list(SS_CASES))
@pytest.mark.parametrize('lg', [True, False])
def test_support_sets_c_oracle_vs_reference(golden, name, lg):
    """The plain-C fp64 restatement (forward + analytic backward) against the reference's autograd."""
    g = golden('support_sets')
    K, N, d, B, seed = SS_CASES[name]
    c = GI.support_sets_case(K, N, d, B, seed, learn_gammas=lg)
    tag = '%s_%s' % (name, 'lg' if lg else 'cg')
    out, _ = O.rbf_c_forward(c['sd'], c['idx'], c['z'], lg, c['gamma'])
    assert rel_err(out, g[tag + '_out']) < 2e-6
    dtable, dal, dlg, dz = O.rbf_c_backward(c['sd'], c['idx'], c['z'], c['gout'], lg, c['gamma'])
    assert rel_err(dz, g[tag + '_dz']) < 2e-5
    assert rel_err(dal, g[tag + '_dalphas']) < 2e-5
    if lg:
        # d/dloggamma sums terms ~ (1 - gamma*r2) with gamma*r2 ~ 1: the reference's own fp32 autograd
        # carries ~1e-4 cancellation noise relative to this fp64 evaluation.
        assert rel_err(dlg.reshape(-1, 1), g[tag + '_dloggamma']) < 5e-4
    if tag + '_dtable' in g:
        assert rel_err(dtable, g[tag + '_dtable']) < 2e-5
    else:
        rows = torch.unique(c['idx']).numpy()
        assert rel_err(dtable[rows][:, ::16], g[tag + '_dtable_rows_sub16']) < 2e-5
        vn = np.linalg.norm(dtable[rows].reshape(len(rows), 2 * N, d), axis=2)
        assert rel_err(vn, g[tag + '_dtable_vecnorm']) < 2e-5
        rest = np.delete(dtable, rows, axis=0)
        assert np.abs(rest).max() == 0.0 and g[tag + '_dtable_rest_absmax'] == 0.0


def test_traverse_oracle_vs_reference(golden):
    g = golden('support_sets')
    c = GI.support_sets_case(6, 3, 16, 2, 21, learn_gammas=True)
    path, _ = O.traverse_paths(c['sd'], c['z'], 0.2, 3, True, c['gamma'])
    assert rel_err(path, g['traverse_path']) < 1e-6


def test_upfirdn2d_oracle_vs_reference_native(golden):
    g = golden('native_ops')
    k = O.make_blur_kernel()
    for i, c in enumerate(GI.UPFIRDN_CASES):
        x = GI.rt(100 + i, c['major'], c['h'], c['w'], c['minor'])
        kk = k * c['gain'] if c['name'] != 'up3_down2' else GI.rt(777, 5, 3)
        y = O.upfirdn2d_mhwc(x, kk, c['up'], c['up'], c['down'], c['down'], *c['pad'])
        ref = g['upfirdn_' + c['name']]
        assert tuple(y.shape) == ref.shape, c['name']
        assert rel_err(y, ref) < 1e-6, c['name']


def _sg2_sd(size, seed):
    """State dict with the reference Generator's keys/shapes, filled from the shared seeded recipe.
    The key/shape manifest comes from the product module (checked against the reference's own
    state_dict by tests/golden/stylegan2.npz's manifest hash in test_stylegan2_manifest)."""
    from warpedganspace_amd.stylegan2 import Generator
    G = Generator(size, 512, 8)
    return GI.fill_state_dict(G.state_dict(), seed)


def test_stylegan2_oracle_vs_reference_g32(golden):
    g = golden('stylegan2')
    sd = _sg2_sd(32, 400 + 32)
    z = GI.rt(410 + 32, 2, 512)
    shift = (GI.rt(411 + 32, 2, 512) * 0.02).requires_grad_(True)
    img = O.sg2_generate(sd, z, 32, shift)
    probe = GI.rt(412 + 32, *img.shape)
    (img * probe).sum().backward()
    assert rel_err(img, g['g32_img']) < 1e-5
    assert rel_err(O.sg2_mapping(sd, z), g['g32_w']) < 1e-5
    assert rel_err(shift.grad, g['g32_dshift']) < 1e-4
    w = O.sg2_mapping(sd, z).detach()
    shw = (GI.rt(413 + 32, 2, 512) * 0.05).requires_grad_(True)
    imgw = O.sg2_synthesis(sd, w + shw, 32)
    (imgw * probe).sum().backward()
    assert rel_err(imgw, g['g32_w_img']) < 1e-5
    assert rel_err(shw.grad, g['g32_w_dshift']) < 1e-4


def test_stylegan2_oracle_vs_reference_g256(golden):
    g = golden('stylegan2')
    sd = _sg2_sd(256, 400 + 256)
    z = GI.rt(410 + 256, 2, 512)
    shift = GI.rt(411 + 256, 2, 512) * 0.02
    with torch.no_grad():
        img = O.sg2_generate(sd, z, 256, shift)
    assert rel_err(torch.nn.functional.avg_pool2d(img, 8), g['g256_img_pool8']) < 1e-5
    assert rel_err(img[:, :, 100:116, 60:76], g['g256_img_crop']) < 1e-5
    assert abs(img.abs().mean().item() - float(g['g256_img_absmean'])) < 1e-5 * float(g['g256_img_absmean'])


@pytest.mark.parametrize('name,cin,cout,ks,up,demod,hw', [('plain', 16, 24, 3, False, True, 8),
                                                            ('up', 16, 8, 3, True, True, 5),
                                                            ('rgb', 16, 3, 1, False, False, 8)])
def test_modulated_conv_oracle_vs_reference(golden, name, cin, cout, ks, up, demod, hw):
    """ModulatedConv2d blocks (model.py:187-228): output, d/dx and d/dstyle against the reference."""
    from collections import OrderedDict
    g = golden('stylegan2')
    tmpl = OrderedDict()
    tmpl['weight'] = torch.zeros(1, cout, cin, ks, ks)
    if up:
        tmpl['blur.kernel'] = O.make_blur_kernel() * 4
    tmpl['modulation.weight'] = torch.zeros(cin, 32)
    tmpl['modulation.bias'] = torch.zeros(cin)
    sd = {'m.' + k: v for k, v in GI.fill_state_dict(tmpl, 300 + len(name)).items()}
    x = GI.rt(310, 2, cin, hw, hw).requires_grad_(True)
    s = GI.rt(311, 2, 32).requires_grad_(True)
    y = O.sg2_modulated_conv(sd, 'm', x, s, demodulate=demod, upsample=up)
    (y * GI.rt(312, *y.shape)).sum().backward()
    assert rel_err(y, g['modconv_%s_y' % name]) < 1e-5
    assert rel_err(x.grad, g['modconv_%s_dx' % name]) < 1e-5
    assert rel_err(s.grad, g['modconv_%s_ds' % name]) < 1e-5


def _proggan_sd(seed, num_blocks=18):
    from warpedganspace_amd.proggan import Generator
    return GI.fill_state_dict(Generator(num_blocks).state_dict(), seed)


def test_proggan_oracle_vs_reference(golden):
    g = golden('generators')
    sd = _proggan_sd(500)
    z = GI.rt(501, 2, 512)
    sh = (GI.rt(502, 2, 512) * 0.1).requires_grad_(True)
    img = O.proggan_generate(sd, z, sh)
    (torch.nn.functional.avg_pool2d(img, 32) * GI.rt(503, 2, 3, 32, 32)).sum().backward()
    assert rel_err(torch.nn.functional.avg_pool2d(img.detach(), 32), g['proggan_img_pool32']) < 1e-5
    assert rel_err(img.detach()[:, :, 500:516, 300:316], g['proggan_img_crop']) < 1e-5
    assert rel_err(sh.grad, g['proggan_dshift']) < 1e-4


def _sngan(tag):
    from warpedganspace_amd.sngan import SN_RES_GEN_CONFIGS, make_resnet_generator
    cfgname, ch, size, seed = {'mnist': ('sn_resnet32', 1, 32, 520), 'anime': ('sn_resnet64', 3, 64, 530)}[tag]
    G = make_resnet_generator(SN_RES_GEN_CONFIGS[cfgname], img_size=size, channels=ch, latent_dim=128)
    G.load_state_dict(GI.fill_state_dict(G.state_dict(), seed, fan_in=True))
    return G, G.state_dict(), SN_RES_GEN_CONFIGS[cfgname].channels, size, seed


@pytest.mark.parametrize('tag', ['mnist', 'anime'])
def test_sngan_oracle_vs_reference(golden, tag):
    g = golden('generators')
    _, sd, channels, size, seed = _sngan(tag)
    z = GI.rt(seed + 1, 3, 128)
    sh = (GI.rt(seed + 2, 3, 128) * 0.1).requires_grad_(True)
    img = O.sngan_generate(sd, z, sh, channels=channels)
    (img * GI.rt(seed + 3, *img.shape)).sum().backward()
    ref = g['sngan_%s_img' % tag]
    assert rel_err(img.detach() if size == 32 else torch.nn.functional.avg_pool2d(img.detach(), 4), ref) < 1e-5
    assert rel_err(sh.grad, g['sngan_%s_dshift' % tag]) < 1e-4


def _lenet(tag):
    from warpedganspace_amd.reconstructor import Reconstructor
    K, c, B, S = {'cfg1': (32, 1, 16, 32), 'rgb': (8, 3, 5, 64)}[tag]
    R = Reconstructor('LeNet', K, channels=c)
    R.load_state_dict(GI.fill_state_dict(R.state_dict(), 700 + K, fan_in=True))
    x1, x2 = GI.rt(701 + K, B, c, S, S), GI.rt(702 + K, B, c, S, S)
    return R, K, B, x1, x2, GI.rt(703 + K, B, K), GI.rt(704 + K, B)


@pytest.mark.parametrize('tag', ['cfg1', 'rgb'])
def test_lenet_oracle_vs_reference(golden, tag):
    g = golden('reconstructor')
    R, K, B, x1, x2, pl, pm = _lenet(tag)
    sd = {k: v.detach().clone().contiguous() for k, v in R.state_dict().items()}
    for k in list(sd):
        if sd[k].is_floating_point() and 'running' not in k:
            sd[k].requires_grad_(True)
    x2 = x2.requires_grad_(True)
    logits, mag = O.reconstructor_lenet(sd, x1, x2, training=True)
    ((logits * pl).sum() + (mag * pm).sum()).backward()
    assert rel_err(logits, g['lenet_%s_logits' % tag]) < 1e-5 and rel_err(mag, g['lenet_%s_mag' % tag]) < 1e-5
    dx2 = x2.grad if tag == 'cfg1' else x2.grad[:, :, ::4, ::4]
    assert rel_err(dx2, g['lenet_%s_dx2' % tag]) < 1e-4
    for n, _ in R.named_parameters():
        gr = sd[n].grad if sd[n].grad.numel() <= 4096 else sd[n].grad.reshape(-1)[::7]
        ref = torch.from_numpy(g['lenet_%s_grad_%s' % (tag, n)])
        assert float((gr - ref).abs().max()) / max(float(ref.abs().max()), 1e-3) < 1e-4, n
    for n, _ in R.named_buffers():
        if 'running' in n:
            assert rel_err(sd[n], g['lenet_%s_buf_%s' % (tag, n)]) < 1e-5, n


def cfg1_setup():
    """BASELINE config 1: SNGAN-MNIST 32x32 generator, LeNet reconstructor, K=32, N=8, B=16 — seeded exactly like
    tools/make_golden.py::gen_step."""
    from warpedganspace_amd.reconstructor import Reconstructor
    from warpedganspace_amd.sngan import SN_RES_GEN_CONFIGS, make_resnet_generator
    K, N, B, d = 32, 8, 16, 128
    Gw = make_resnet_generator(SN_RES_GEN_CONFIGS['sn_resnet32'], img_size=32, channels=1, latent_dim=d)
    Gw.load_state_dict(GI.fill_state_dict(Gw.state_dict(), 800, fan_in=True))
    c = GI.support_sets_case(K, N, d, B, 801, learn_gammas=True)
    R = Reconstructor('LeNet', K, channels=1)
    R.load_state_dict(GI.fill_state_dict(R.state_dict(), 802, fan_in=True))
    mag = GI.rt(803, B).abs() * 0.1 + 0.25
    mag = mag * torch.where(GI.rt(804, B) > -0.5, 1.0, -1.0)
    return Gw, c, R, mag, (K, N, B, d)


def test_cfg1_training_step_oracle_vs_reference(golden):
    """The oracle's replay of lib/trainer.py:190-254 against the same step driven through the reference modules."""
    g = golden('step_cfg1')
    Gw, c, R, mag, (K, N, B, d) = cfg1_setup()
    sd_r = {k: v.detach().clone().contiguous() for k, v in R.state_dict().items()}
    ref = O.ReferenceStep(Gw.state_dict(), c['sd'], sd_r, 32, learn_gammas=True, gamma=c['gamma'], reconstructor='LeNet',
                          generator='SNGAN', gen_kwargs=dict(channels=(256, 256, 256, 256)), g_requires_grad=True)
    o = ref.step(c['z'], c['idx'], mag)
    assert abs(o['loss'] - float(g['step_loss'])) < 1e-5 and abs(o['ce'] - float(g['step_ce'])) < 1e-5
    assert abs(o['l1'] - float(g['step_l1'])) < 1e-5 and abs(o['acc'] - float(g['step_acc'])) < 1e-7
    assert torch.equal(o['argmax'], torch.from_numpy(g['step_argmax']))
    assert rel_err(o['shift'], g['step_shift']) < 1e-5 and rel_err(o['logits'], g['step_logits']) < 1e-4
    rows = torch.unique(c['idx'])
    assert rel_err(ref.s['SUPPORT_SETS'].grad[rows][:, ::4], g['step_dS_rows']) < 1e-3
    assert rel_err(ref.s['LOGGAMMA'].detach(), g['step_post_loggamma']) < 1e-6


def _biggan():
    from warpedganspace_amd.biggan import build_biggan
    W = build_biggan(None, (239,))
    W.G.load_state_dict(GI.fill_state_dict(W.G.state_dict(), 540, fan_in=True, per_key=True))
    return W


def test_biggan_manifest_and_oracle_vs_reference(golden):
    g = golden('generators')
    W = _biggan()
    sd = W.G.state_dict()
    assert sorted(sd.keys()) == list(g['biggan_keys'])                     # same state_dict manifest as the reference
    z = GI.rt(541, 2, 120)
    sh = (GI.rt(542, 2, 120) * 0.1).requires_grad_(True)
    img = O.biggan_generate(sd, z, torch.tensor([239, 100]), sh)
    (img * GI.rt(543, *img.shape)).sum().backward()
    assert rel_err(torch.nn.functional.avg_pool2d(img.detach(), 4), g['biggan_img_pool4']) < 1e-5
    assert rel_err(img.detach()[:, :, 40:56, 70:86], g['biggan_img_crop']) < 1e-5
    assert rel_err(sh.grad, g['biggan_dshift']) < 1e-4
