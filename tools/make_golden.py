#!/usr/bin/env python
"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE (build container only).

Works from a scratch copy of /root/reference (never writes into it), with bytecode disabled, and
with `models.StyleGAN2.op` pre-stubbed so importing models/StyleGAN2/model.py does not JIT-build
(hipify) the reference's CUDA extension (SURVEY.md hazard note).  The stub's `upfirdn2d` is the
reference's OWN pure-PyTorch statement of the op, `upfirdn2d_native` (models/StyleGAN2/op/
upfirdn2d.py:152-186), extracted with `ast` from the copied file and executed with `F` injected
(the reference forgot the import); `fused_leaky_relu` is restated from fused_bias_act_kernel.cu:25-47.

Nothing from the reference is stored: only numeric outputs for seeded inputs (tests/golden_inputs.py).
Usage:  python tools/make_golden.py [--only support_sets,native_ops,stylegan2,...]
"""
import argparse
import ast
import os
import shutil
import sys
import tempfile
import types

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests import golden_inputs as GI  # noqa: E402

REF = '/root/reference'
GOLD = os.path.join(REPO, 'tests', 'golden')


def stage_reference():
    tmp = tempfile.mkdtemp(prefix='refcopy_')
    for sub in ('lib', 'models'):
        shutil.copytree(os.path.join(REF, sub), os.path.join(tmp, sub),
                        ignore=shutil.ignore_patterns('__pycache__', '*.pyc', 'pretrained'))
    sys.path.insert(0, tmp)
    return tmp


def load_module_from(path, name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reference_upfirdn2d_native(tmp):
    src = open(os.path.join(tmp, 'models', 'StyleGAN2', 'op', 'upfirdn2d.py')).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'upfirdn2d_native'][0]
    ns = {'torch': torch, 'F': F}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'upfirdn2d_native', 'exec'), ns)
    return ns['upfirdn2d_native']


def install_stylegan2_op_stub(tmp):
    native = reference_upfirdn2d_native(tmp)

    def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
        b, c, h, w = x.shape
        out = native(x.reshape(-1, h, w, 1), kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
        return out.reshape(b, c, out.shape[1], out.shape[2])

    def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
        return F.leaky_relu(x + bias.view(1, -1, *([1] * (x.ndim - 2))), negative_slope) * scale

    class FusedLeakyReLU(torch.nn.Module):
        def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
            super().__init__()
            self.bias = torch.nn.Parameter(torch.zeros(channel))
            self.negative_slope, self.scale = negative_slope, scale

        def forward(self, x):
            return fused_leaky_relu(x, self.bias, self.negative_slope, self.scale)

    stub = types.ModuleType('models.StyleGAN2.op')
    stub.upfirdn2d, stub.fused_leaky_relu, stub.FusedLeakyReLU = upfirdn2d, fused_leaky_relu, FusedLeakyReLU
    sys.modules['models.StyleGAN2.op'] = stub
    return native


# ------------------------------------------------------------------------------------------------
def gen_support_sets(tmp):
    ss = load_module_from(os.path.join(tmp, 'lib', 'support_sets.py'), 'ref_support_sets')
    out = {}
    cases = {'tiny': (4, 2, 8, 3, 11), 'cfg1': (32, 8, 128, 4, 12), 'cfg3': (128, 32, 512, 4, 13),
             'cfg4': (16, 4, 120, 5, 14)}
    for name, (K, N, d, B, seed) in cases.items():
        for lg in (True, False):
            c = GI.support_sets_case(K, N, d, B, seed, learn_gammas=lg)
            m = ss.SupportSets(K, N, d, learn_alphas=True, learn_gammas=lg, gamma=c['gamma'])
            m.load_state_dict(c['sd'])
            z = c['z'].clone().requires_grad_(True)
            y = m(GI.one_hot(c['idx'], K), z)
            (y * c['gout']).sum().backward()
            tag = '%s_%s' % (name, 'lg' if lg else 'cg')
            out[tag + '_out'] = y.detach().numpy()
            out[tag + '_dz'] = z.grad.numpy()
            out[tag + '_dalphas'] = m.ALPHAS.grad.numpy()
            if lg:
                out[tag + '_dloggamma'] = m.LOGGAMMA.grad.numpy()
            g = m.SUPPORT_SETS.grad
            if name in ('tiny', 'cfg4'):
                out[tag + '_dtable'] = g.numpy()
            else:  # store only the selected rows (all others are exactly zero in the reference too)
                rows = torch.unique(c['idx'])
                gr = g[rows]
                out[tag + '_dtable_rows_sub16'] = gr[:, ::16].numpy()
                out[tag + '_dtable_vecnorm'] = gr.reshape(len(rows), 2 * N, d).norm(dim=2).numpy()
                out[tag + '_dtable_rest_absmax'] = np.float32(
                    g[[k for k in range(K) if k not in set(rows.tolist())]].abs().max().item())
    # traversal (traverse_latent_space.py:361-438), K=6, T=3
    c = GI.support_sets_case(6, 3, 16, 2, 21, learn_gammas=True)
    m = ss.SupportSets(6, 3, 16, learn_alphas=False, learn_gammas=True, gamma=c['gamma'])
    m.load_state_dict(c['sd'])
    eps, T = 0.2, 3
    path = np.zeros((2, 6, 2 * T + 1, 16), np.float32)
    with torch.no_grad():
        for ci in range(2):
            for k in range(6):
                mask = torch.zeros(1, 6)
                mask[0, k] = 1.0
                path[ci, k, T] = c['z'][ci].numpy()
                for sign in (1.0, -1.0):
                    zc = c['z'][ci:ci + 1].clone()
                    for t in range(1, T + 1):
                        zc = zc + sign * eps * m(mask, zc)
                        path[ci, k, T + int(sign) * t] = zc[0].numpy()
    out['traverse_path'] = path
    np.savez_compressed(os.path.join(GOLD, 'support_sets.npz'), **out)
    print('support_sets.npz', len(out), 'arrays')


def gen_native_ops(tmp, native):
    out = {}
    k1 = torch.tensor([1., 3., 3., 1.])
    k = k1[None, :] * k1[:, None]
    k = k / k.sum()
    for i, c in enumerate(GI.UPFIRDN_CASES):
        x = GI.rt(100 + i, c['major'], c['h'], c['w'], c['minor'])
        kk = k * c['gain']
        if c['name'] == 'up3_down2':
            kk = GI.rt(777, 5, 3)
        y = native(x, kk, c['up'], c['up'], c['down'], c['down'], *c['pad'])
        out['upfirdn_' + c['name']] = y.numpy()
    np.savez_compressed(os.path.join(GOLD, 'native_ops.npz'), **out)
    print('native_ops.npz', len(out), 'arrays')


def gen_stylegan2(tmp):
    from models.StyleGAN2.model import Generator, ModulatedConv2d
    out = {}
    torch.manual_seed(0)
    # --- ModulatedConv2d blocks (resolution-agnostic)
    for name, (cin, cout, ks, up, demod, hw) in {
            'plain': (16, 24, 3, False, True, 8), 'up': (16, 8, 3, True, True, 5),
            'rgb': (16, 3, 1, False, False, 8)}.items():
        m = ModulatedConv2d(cin, cout, ks, 32, demodulate=demod, upsample=up)
        m.load_state_dict(GI.fill_state_dict(m.state_dict(), 300 + len(name)))
        x = GI.rt(310, 2, cin, hw, hw).requires_grad_(True)
        s = GI.rt(311, 2, 32).requires_grad_(True)
        y = m(x, s)
        probe = GI.rt(312, *y.shape)
        (y * probe).sum().backward()
        out['modconv_%s_y' % name] = y.detach().numpy()
        out['modconv_%s_dx' % name] = x.grad.numpy()
        out['modconv_%s_ds' % name] = s.grad.numpy()
    # --- full generators, Z-space forward + gradient w.r.t. the shift
    for size, B in ((32, 2), (256, 2)):
        G = Generator(size, 512, 8)
        G.load_state_dict(GI.fill_state_dict(G.state_dict(), 400 + size))
        G.eval()
        z = GI.rt(410 + size, B, 512)
        shift = (GI.rt(411 + size, B, 512) * 0.02).requires_grad_(True)
        img = G([z + shift], input_is_latent=False)[0]
        probe = GI.rt(412 + size, *img.shape)
        (img * probe).sum().backward()
        w = G.get_latent(z).detach()
        tag = 'g%d_' % size
        if size == 32:
            out[tag + 'img'] = img.detach().numpy()
        else:
            out[tag + 'img_pool8'] = F.avg_pool2d(img.detach(), 8).numpy()
            out[tag + 'img_crop'] = img.detach()[:, :, 100:116, 60:76].numpy()
            out[tag + 'img_absmean'] = np.float32(img.detach().abs().mean().item())
        out[tag + 'w'] = w.numpy()
        out[tag + 'dshift'] = shift.grad.numpy()
        # W-space: image from w + shift and gradient w.r.t. that shift
        shw = (GI.rt(413 + size, B, 512) * 0.05).requires_grad_(True)
        imgw = G([w + shw], input_is_latent=True)[0]
        (imgw * probe).sum().backward()
        out[tag + 'w_dshift'] = shw.grad.numpy()
        if size == 32:
            out[tag + 'w_img'] = imgw.detach().numpy()
        else:
            out[tag + 'w_img_pool8'] = F.avg_pool2d(imgw.detach(), 8).numpy()
        # the same gradients from the reference modules run in float64: the fp32 reference itself is
        # ~1e-3 away from these (ill-conditioned random mapping net), so tests measure against fp64
        G64 = G.double()
        sh64 = shift.detach().double().requires_grad_(True)
        (G64([z.double() + sh64], input_is_latent=False)[0] * probe.double()).sum().backward()
        out[tag + 'dshift64'] = sh64.grad.numpy()
        shw64 = shw.detach().double().requires_grad_(True)
        w64 = G64.get_latent(z.double()).detach()
        (G64([w64 + shw64], input_is_latent=True)[0] * probe.double()).sum().backward()
        out[tag + 'w_dshift64'] = shw64.grad.numpy()
    np.savez_compressed(os.path.join(GOLD, 'stylegan2.npz'), **out)
    print('stylegan2.npz', len(out), 'arrays')


def gen_generators(tmp):
    """ProgGAN (models/ProgGAN/model.py) — full 1024^2 network, B=2: pooled image, 8-block prefix, d/dz."""
    from models.ProgGAN.model import Generator as PG
    out = {}
    G = PG()
    G.load_state_dict(GI.fill_state_dict(G.state_dict(), 500))
    G.eval()
    z = GI.rt(501, 2, 512)
    sh = (GI.rt(502, 2, 512) * 0.1).requires_grad_(True)
    x = (z + sh).reshape(2, 512, 1, 1)
    feat8 = G.features[:8](x)
    out['proggan_feat8_sub'] = feat8.detach()[:, ::8, ::2, ::2].numpy()    # [2,64,16,16] sub-sample of [2,512,32,32]
    img = G.output(G.features[8:](feat8))
    probe = GI.rt(503, 2, 3, 32, 32)
    (F.avg_pool2d(img, 32) * probe).sum().backward()
    out['proggan_img_pool32'] = F.avg_pool2d(img.detach(), 32).numpy()
    out['proggan_img_crop'] = img.detach()[:, :, 500:516, 300:316].numpy()
    out['proggan_dshift'] = sh.grad.numpy()
    # SNGAN (models/SNGAN/sn_gen_resnet.py), eval mode, both configurations used by build_sngan
    from models.SNGAN.sn_gen_resnet import SN_RES_GEN_CONFIGS, make_resnet_generator
    from models.SNGAN.distribution import NormalDistribution
    for tag, cfgname, ch, size, seed in (('mnist', 'sn_resnet32', 1, 32, 520), ('anime', 'sn_resnet64', 3, 64, 530)):
        Gs = make_resnet_generator(SN_RES_GEN_CONFIGS[cfgname], img_size=size, channels=ch, distribution=NormalDistribution(128))
        Gs.load_state_dict(GI.fill_state_dict(Gs.state_dict(), seed, fan_in=True))
        Gs.eval()
        z = GI.rt(seed + 1, 3, 128)
        sh = (GI.rt(seed + 2, 3, 128) * 0.1).requires_grad_(True)
        img = Gs.model(z + sh)
        probe = GI.rt(seed + 3, *img.shape)
        (img * probe).sum().backward()
        out['sngan_%s_img' % tag] = img.detach().numpy() if size == 32 else F.avg_pool2d(img.detach(), 4).numpy()
        out['sngan_%s_dshift' % tag] = sh.grad.numpy()
    # BigGAN-128 reference architecture (models/BigGAN/BigGAN.py + generator_config.json), eval mode
    import json
    from models.BigGAN import BigGAN as BG, utils as BU
    with open(os.path.join(tmp, 'models', 'BigGAN', 'generator_config.json')) as f:
        config = json.load(f)
    config['resolution'] = BU.imsize_dict[config['dataset']]
    config['n_classes'] = BU.nclass_dict[config['dataset']]
    config['G_activation'] = BU.activation_dict[config['G_nl']]
    config['D_activation'] = BU.activation_dict[config['D_nl']]
    config['skip_init'] = True
    config['no_optim'] = True
    Gb = BG.Generator(**config)
    Gb.load_state_dict(GI.fill_state_dict(Gb.state_dict(), 540, fan_in=True, per_key=True))
    Gb.eval()
    out['biggan_keys'] = np.array(sorted(Gb.state_dict().keys()))
    z = GI.rt(541, 2, Gb.dim_z)
    sh = (GI.rt(542, 2, Gb.dim_z) * 0.1).requires_grad_(True)
    yb = Gb.shared(torch.tensor([239, 100]))
    img = Gb(z + sh, yb)
    probe = GI.rt(543, *img.shape)
    (img * probe).sum().backward()
    out['biggan_img_pool4'] = F.avg_pool2d(img.detach(), 4).numpy()
    out['biggan_img_crop'] = img.detach()[:, :, 40:56, 70:86].numpy()
    out['biggan_dshift'] = sh.grad.numpy()
    np.savez_compressed(os.path.join(GOLD, 'generators.npz'), **out)
    print('generators.npz', len(out), 'arrays')


def gen_reconstructor(tmp):
    """LeNet reconstructor (lib/reconstructor.py:18-49,72-75), train mode, cfg1 shape: outputs, every
    parameter gradient, d/dx2 and the updated running statistics.  (torchvision is absent: a stub module only
    satisfies `from torchvision.models import resnet18`; the ResNet branch cannot be pinned here.)"""
    tv, tvm = types.ModuleType('torchvision'), types.ModuleType('torchvision.models')
    tvm.resnet18 = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('torchvision is not available'))
    tv.models = tvm
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.models', tvm)
    rec = load_module_from(os.path.join(tmp, 'lib', 'reconstructor.py'), 'ref_reconstructor')
    out = {}
    for tag, (K, c, B, S) in {'cfg1': (32, 1, 16, 32), 'rgb': (8, 3, 5, 64)}.items():
        R = rec.Reconstructor('LeNet', K, channels=c)
        R.load_state_dict(GI.fill_state_dict(R.state_dict(), 700 + K, fan_in=True))
        R.train()
        x1 = GI.rt(701 + K, B, c, S, S)
        x2 = GI.rt(702 + K, B, c, S, S).requires_grad_(True)
        logits, mag = R(x1, x2)
        (logits * GI.rt(703 + K, B, K)).sum().add((mag * GI.rt(704 + K, B)).sum()).backward()
        out['lenet_%s_logits' % tag] = logits.detach().numpy()
        out['lenet_%s_mag' % tag] = mag.detach().numpy()
        out['lenet_%s_dx2' % tag] = x2.grad.numpy() if tag == 'cfg1' else x2.grad[:, :, ::4, ::4].numpy()
        for n, p in R.named_parameters():
            gr = p.grad
            out['lenet_%s_grad_%s' % (tag, n)] = (gr if gr.numel() <= 4096 else gr.reshape(-1)[::7]).numpy()
        for n, b in R.named_buffers():
            if 'running' in n:
                out['lenet_%s_buf_%s' % (tag, n)] = b.numpy()
    np.savez_compressed(os.path.join(GOLD, 'reconstructor.npz'), **out)
    print('reconstructor.npz', len(out), 'arrays')


def gen_step(tmp):
    """One optimisation step of BASELINE config 1 (SNGAN-MNIST 32x32, LeNet, K=32, N=8, B=16) driven through the
    REFERENCE modules (SNGANWrapper-equivalent model, SupportSets, Reconstructor) with the loop body of
    lib/trainer.py:190-254 restated (lib/trainer.py itself needs tensorboard) and torch.optim.Adam."""
    tv, tvm = types.ModuleType('torchvision'), types.ModuleType('torchvision.models')
    tvm.resnet18 = lambda *a, **k: None
    tv.models = tvm
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.models', tvm)
    rec = load_module_from(os.path.join(tmp, 'lib', 'reconstructor.py'), 'ref_reconstructor_step')
    ss = load_module_from(os.path.join(tmp, 'lib', 'support_sets.py'), 'ref_support_sets2')
    from models.SNGAN.sn_gen_resnet import SN_RES_GEN_CONFIGS, make_resnet_generator
    from models.SNGAN.distribution import NormalDistribution
    K, N, B, d = 32, 8, 16, 128
    Gw = make_resnet_generator(SN_RES_GEN_CONFIGS['sn_resnet32'], img_size=32, channels=1, distribution=NormalDistribution(d))
    Gw.load_state_dict(GI.fill_state_dict(Gw.state_dict(), 800, fan_in=True))
    G = Gw.model
    c = GI.support_sets_case(K, N, d, B, 801, learn_gammas=True)
    S = ss.SupportSets(K, N, d, learn_alphas=False, learn_gammas=True, gamma=c['gamma'])
    S.load_state_dict(c['sd'])
    R = rec.Reconstructor('LeNet', K, channels=1)
    R.load_state_dict(GI.fill_state_dict(R.state_dict(), 802, fan_in=True))
    G.eval(); S.train(); R.train()
    opt_s = torch.optim.Adam(S.parameters(), lr=1e-4)
    opt_r = torch.optim.Adam(R.parameters(), lr=1e-4)
    z, idx = c['z'], c['idx']
    mag = GI.rt(803, B).abs() * 0.1 + 0.25
    mag = mag * torch.where(GI.rt(804, B) > -0.5, 1.0, -1.0)
    G.zero_grad(); S.zero_grad(); R.zero_grad()
    img = G(z)
    mask = torch.zeros(B, K)
    for i, index in enumerate(idx):
        mask[i][index] += 1.0
    shift = mag.reshape(-1, 1) * S(mask, z)
    img_shifted = G(z + shift)
    logits, mag_hat = R(img, img_shifted)
    ce = torch.nn.CrossEntropyLoss()(logits, idx)
    l1 = torch.mean(torch.abs(mag_hat - mag))
    loss = 1.0 * ce + 0.25 * l1
    loss.backward()
    out = {'step_loss': np.float32(loss.item()), 'step_ce': np.float32(ce.item()), 'step_l1': np.float32(l1.item()),
           'step_acc': np.float32((logits.argmax(1) == idx).float().mean().item()), 'step_argmax': logits.argmax(1).numpy(),
           'step_logits': logits.detach().numpy(), 'step_shift': shift.detach().numpy(),
           'step_img_shifted': img_shifted.detach().numpy()[:4],
           'step_dS_rows': S.SUPPORT_SETS.grad[torch.unique(idx)].numpy()[:, ::4], 'step_dloggamma': S.LOGGAMMA.grad.numpy()}
    for n, p in R.named_parameters():
        out['step_gradnorm_' + n] = np.float32(p.grad.norm().item())
    opt_s.step(); opt_r.step()
    out['step_post_loggamma'] = S.LOGGAMMA.detach().numpy()
    out['step_post_S_absmean_update'] = np.float32((S.SUPPORT_SETS.detach() - c['sd']['SUPPORT_SETS']).abs().mean().item())
    np.savez_compressed(os.path.join(GOLD, 'step_cfg1.npz'), **out)
    print('step_cfg1.npz', len(out), 'arrays')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    only = set(filter(None, args.only.split(',')))
    os.makedirs(GOLD, exist_ok=True)
    tmp = stage_reference()
    try:
        native = install_stylegan2_op_stub(tmp)
        gens = {'support_sets': lambda: gen_support_sets(tmp), 'native_ops': lambda: gen_native_ops(tmp, native),
                'stylegan2': lambda: gen_stylegan2(tmp)}
        for extra in ('gen_reconstructor', 'gen_generators', 'gen_step'):
            if extra in globals():
                gens[extra[4:]] = (lambda f: (lambda: f(tmp)))(globals()[extra])
        for name, fn in gens.items():
            if not only or name in only:
                fn()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
