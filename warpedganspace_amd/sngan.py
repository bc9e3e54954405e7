"""SNGAN ResNet generator on the HIP kernels — host-side mirror of models/SNGAN/sn_gen_resnet.py:9-112
and SNGANWrapper / build_sngan (models/gan_load.py:21-57).  The module tree reproduces the reference's
state_dict keys (`model.0.*` Linear, `model.{2..}.{conv1,conv2,model.0,model.3,model.4,model.6,bypass.1}.*`,
final BN `model.{n}.*`, final conv `model.{n+2}.*`), including the reference's duplicate entries
(`conv1` is the same tensor as `model.3`).

Generator in eval mode (lib/trainer.py:143-150): BatchNorm = per-channel affine from the running statistics
(fused with ReLU), nearest up-sampling folded into the conv gather, bias / bypass add / tanh in the GEMM
epilogue.  Backward propagates only the input gradient.
"""
from collections import namedtuple

import functools

import numpy as np
import torch
from torch import nn

from . import _lib as L
from . import conv as C

ResNetGenConfig = namedtuple('ResNetGenConfig', ['channels', 'seed_dim'])
SN_RES_GEN_CONFIGS = {
    'sn_resnet32': ResNetGenConfig([256, 256, 256, 256], 4),
    'sn_resnet64': ResNetGenConfig([16 * 64, 8 * 64, 4 * 64, 2 * 64, 64], 4),
}
SNGAN_CONFIG = {
    'SNGAN_MNIST': {'image_channels': 1, 'latent_dim': 128, 'model': 'sn_resnet32', 'img_size': 32},
    'SNGAN_AnimeFaces': {'image_channels': 3, 'latent_dim': 128, 'model': 'sn_resnet64', 'img_size': 64},
}


class _Marker(nn.Module):
    """Parameter-free placeholder keeping the reference's Sequential indices (Reshape / ReLU / Upsample / Tanh)."""


class ResBlockGenerator(nn.Module):
    """Parameter container of sn_gen_resnet.py:24-54 (same attribute layout => same state_dict keys)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, padding=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, padding=1)
        nn.init.xavier_uniform_(self.conv1.weight.data, np.sqrt(2))
        nn.init.xavier_uniform_(self.conv2.weight.data, np.sqrt(2))
        self.model = nn.Sequential(nn.BatchNorm2d(in_channels), _Marker(), _Marker(), self.conv1,
                                   nn.BatchNorm2d(out_channels), _Marker(), self.conv2)
        if in_channels == out_channels:
            self.bypass = _Marker()
        else:
            self.bypass = nn.Sequential(_Marker(), nn.Conv2d(in_channels, out_channels, 3, 1, padding=1))
            nn.init.xavier_uniform_(self.bypass[1].weight.data, 1.0)
        self.in_channels, self.out_channels = in_channels, out_channels


class _SN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, G, z, prec):
        ctx.prec = prec
        img, saved = G._fwd(z, ctx.needs_input_grad[1], prec)
        ctx.G, ctx.saved = G, saved
        if G.debug_keep is not None and saved is not None:     # ReLU gates in execution order, NCHW (tests)
            gates = []
            for (_, a1, _, _, a2, _) in saved[0]:
                gates += [(a1 > 0).permute(0, 3, 1, 2), (a2 > 0).permute(0, 3, 1, 2)]
            G.debug_keep['gates'] = gates + [(saved[2] > 0).permute(0, 3, 1, 2)]
        return img

    @staticmethod
    def backward(ctx, gimg):
        return None, ctx.G._bwd(ctx.saved, gimg.contiguous(), ctx.prec), None


class GenModel(nn.Module):
    """`GenWrapper.model` of the reference: Sequential(Linear, Reshape, ResBlocks..., BN, ReLU, conv, Tanh)."""

    def __init__(self, cfg, channels, latent_dim):
        super().__init__()
        ch = cfg.channels
        self.seed_dim, self.latent_dim, self.img_channels = cfg.seed_dim, latent_dim, channels
        dense = nn.Linear(latent_dim, cfg.seed_dim ** 2 * ch[0])
        nn.init.xavier_uniform_(dense.weight.data, 1.)
        final = nn.Conv2d(ch[-1], channels, 3, stride=1, padding=1)
        nn.init.xavier_uniform_(final.weight.data, 1.)
        mods = [dense, _Marker()] + [ResBlockGenerator(ch[i], ch[i + 1]) for i in range(len(ch) - 1)] + \
               [nn.BatchNorm2d(ch[-1]), _Marker(), final, _Marker()]
        self.seq = nn.Sequential(*mods)
        self.nblocks = len(ch) - 1
        for p in self.parameters():
            p.requires_grad_(False)
        self._prep = None
        self.debug_keep = None
        self.precision = 'fp32'      # arithmetic of the convs when forward() is not told otherwise (conv.PRECISION_NAMES)

    # expose the Sequential's children under their numeric names: state_dict keys '0.weight', '2.conv1.weight', ...
    def state_dict(self, *a, **k):
        return self.seq.state_dict(*a, **k)

    def load_state_dict(self, sd, strict=True):
        self._prep = None
        return self.seq.load_state_dict(sd, strict=strict)

    def _apply(self, fn, *a, **k):
        self._prep = None
        return super()._apply(fn, *a, **k)

    def _prepare(self):
        dev = self.seq[0].weight.device
        if self._prep is not None and self._prep['dev'] == dev:
            return self._prep
        if dev.type != 'cuda':
            raise L.WgsError("SNGAN generator runs on the HIP kernels only: move it to the GPU (no CPU fallback)")
        s2 = self.seed_dim ** 2
        with torch.no_grad():
            lin = self.seq[0]
            c0 = lin.weight.shape[0] // s2
            # Linear rows re-ordered from (c, h, w) to (h, w, c): its output is then directly the NHWC seed tensor
            perm = torch.arange(c0 * s2, device=dev).reshape(c0, s2).t().reshape(-1)
            P = {'dev': dev, 'lin_w': lin.weight[perm].contiguous(), 'lin_b': lin.bias[perm].contiguous(), 'c0': c0, 'blocks': []}

            def packs(conv, pad_co=None):
                co, ci = conv.weight.shape[:2]
                wp = C.pack_weight(conv.weight.float())
                b = conv.bias.detach().clone()
                if pad_co:
                    wpp = torch.zeros(pad_co, 9, ci, device=dev)
                    wpp[:co] = wp
                    bb = torch.zeros(pad_co, device=dev)
                    bb[:co] = b
                    wp, b, co = wpp, bb, pad_co
                wt = C.repack_w_t(wp, co, 9, ci)
                return dict(wp=wp, wt=wt, wc=C.WinoCache(wp), wtc=C.WinoCache(wt), b=b, ci=ci, co=co)
            for i in range(self.nblocks):
                blk = self.seq[2 + i]
                d = dict(bn1=blk.model[0], bn2=blk.model[4], c1=packs(blk.conv1), c2=packs(blk.conv2),
                         byp=(packs(blk.bypass[1]) if isinstance(blk.bypass, nn.Sequential) else None))
                P['blocks'].append(d)
            n = 2 + self.nblocks
            P['bn'] = self.seq[n]
            P['final'] = packs(self.seq[n + 2], pad_co=8)
            P["ws"] = torch.zeros(64 * 1024, dtype=torch.float64, device=dev)    # WGS_BN_WS_DOUBLES(1024)
        self._prep = P
        return P

    @staticmethod
    def _bn_relu(bn, x, ws):
        """eval-mode BatchNorm (running statistics) + ReLU."""
        N, Cn = x.numel() // x.shape[-1], x.shape[-1]
        y = torch.empty_like(x)
        mean, invstd = torch.empty(Cn, device=x.device), torch.empty(Cn, device=x.device)
        L.check(L.lib().wgs_bn_fwd(L.ptr(x), L.ptr(bn.weight), L.ptr(bn.bias), None, L.ptr(y), L.ptr(mean), L.ptr(invstd),
                                   L.ptr(bn.running_mean), L.ptr(bn.running_var), None, L.rawptr(ws), L.c_int64(N), Cn,
                                   L.c_float(bn.eps), L.c_float(0.1), 1, 0, L.stream()), 'bn_eval')
        return y, (mean, invstd)

    @staticmethod
    def _bn_relu_bwd(bn, x, stats, g, out, ws):
        N, Cn = x.numel() // x.shape[-1], x.shape[-1]
        dx = torch.empty_like(x)
        L.check(L.lib().wgs_bn_bwd(L.ptr(x), L.ptr(g), None, L.ptr(out), L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(bn.weight),
                                   L.ptr(dx), None, None, None, L.rawptr(ws), L.c_int64(N), Cn, 0, L.stream()), 'bn_eval_bwd')
        return dx

    def _fwd(self, z, save, prec):
        P = self._prepare()
        launch = functools.partial(C.launch, precision=prec)
        lib, st = L.lib(), L.stream()
        z = z.contiguous()
        B, dz = z.shape
        dev = z.device
        s, c0 = self.seed_dim, P['c0']
        x = torch.empty(B, s, s, c0, device=dev)
        # Linear: K = latent_dim (128), N = s*s*c0 rows already in NHWC order
        L.check(lib.wgs_linear_fwd(L.ptr(z), L.ptr(P['lin_w']), L.ptr(P['lin_b']), L.ptr(x), B, s * s * c0, dz, dz, s * s * c0,
                                   L.c_float(1.0), L.c_float(1.0), 0, 0, L.c_float(0.0), L.c_float(1.0), st), 'seed_linear')
        saved = []
        taps = [(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
        for d in P['blocks']:
            H = x.shape[1]
            a1, s1 = self._bn_relu(d['bn1'], x, P['ws'])
            c1 = d['c1']
            h1 = torch.empty(B, 2 * H, 2 * H, c1['co'], device=dev)
            launch(a1, c1['wp'], h1, taps, 2 * H, 2 * H, w_tap_stride=c1['ci'], w_row_stride=9 * c1['ci'], ups=1, bias=c1['b'])
            a2, s2 = self._bn_relu(d['bn2'], h1, P['ws'])
            c2 = d['c2']
            if d['byp'] is None:
                addend, add_ups = x, 1                       # bypass = nearest up-sample of the block input
            else:
                bp = d['byp']
                addend = torch.empty(B, 2 * H, 2 * H, bp['co'], device=dev)
                launch(x, bp['wp'], addend, taps, 2 * H, 2 * H, w_tap_stride=bp['ci'], w_row_stride=9 * bp['ci'], ups=1, bias=bp['b'])
                add_ups = 0
            y = torch.empty(B, 2 * H, 2 * H, c2['co'], device=dev)
            launch(a2, c2['wp'], y, taps, 2 * H, 2 * H, w_tap_stride=c2['ci'], w_row_stride=9 * c2['ci'], bias=c2['b'],
                     addend=addend, add_ups=add_ups)
            if save:
                saved.append((x, a1, s1, h1, a2, s2))
            x = y
        af, sf = self._bn_relu(P['bn'], x, P['ws'])
        f = P['final']
        Hc = x.shape[1]
        y8 = torch.empty(B, Hc, Hc, 8, device=dev)
        launch(af, f['wp'], y8, taps, Hc, Hc, w_tap_stride=f['ci'], w_row_stride=9 * f['ci'], bias=f['b'], act=1)
        img = y8[..., :self.img_channels].permute(0, 3, 1, 2).contiguous()
        return img, ((saved, x, af, sf, y8, z) if save else None)

    def _bwd(self, saved_all, gimg, prec):
        P = self._prepare()
        # fp16 modes: a gradient operand without a magnitude bound runs in split-bf16 (grad_operand)
        launch = functools.partial(C.launch, precision=prec, grad_operand=True)
        lib, st = L.lib(), L.stream()
        saved, x_last, af, sf, y8, z = saved_all
        B = gimg.shape[0]
        dev = gimg.device
        Hc = gimg.shape[2]
        g8 = torch.zeros(B, Hc, Hc, 8, device=dev)
        g8[..., :self.img_channels] = gimg.permute(0, 2, 3, 1)
        dpre = torch.empty_like(g8)
        L.check(lib.wgs_bias_act(L.ptr(g8), None, L.ptr(y8), L.ptr(dpre), 9, 1, L.c_float(0.0), L.c_float(1.0),
                                 L.c_int64(g8.numel()), 1, 1, st), 'tanh_bwd')
        f = P['final']
        dtaps = [(1 - ky, 1 - kx, ky * 3 + kx) for ky in range(3) for kx in range(3)]
        gaf = torch.empty_like(af)
        launch(dpre, f['wt'], gaf, dtaps, Hc, Hc, w_tap_stride=f['ci'] * 8, w_row_stride=8)
        g = self._bn_relu_bwd(P['bn'], x_last, sf, gaf, af, P['ws'])
        for d, (x, a1, s1, h1, a2, s2) in zip(reversed(P['blocks']), reversed(saved)):
            H = x.shape[1]
            c1, c2 = d['c1'], d['c2']
            # y = conv2(a2) + b2 + bypass(x)
            ga2 = torch.empty_like(a2)
            launch(g, c2['wt'], ga2, dtaps, 2 * H, 2 * H, w_tap_stride=c2['ci'] * c2['co'], w_row_stride=c2['co'], w_split=c2['wtc'])
            gh1 = self._bn_relu_bwd(d['bn2'], h1, s2, ga2, a2, P['ws'])
            gup = torch.empty(B, 2 * H, 2 * H, c1['ci'], device=dev)
            launch(gh1, c1['wt'], gup, dtaps, 2 * H, 2 * H, w_tap_stride=c1['ci'] * c1['co'], w_row_stride=c1['co'], w_split=c1['wtc'])
            ga1 = torch.empty_like(a1)
            L.check(lib.wgs_upsample2x_bwd(L.ptr(gup), L.ptr(ga1), B, H, H, c1['ci'], st), 'up_bwd')
            gx = self._bn_relu_bwd(d['bn1'], x, s1, ga1, a1, P['ws'])
            if d['byp'] is None:
                gb_up = g
            else:
                bp = d['byp']
                gb_up = torch.empty(B, 2 * H, 2 * H, bp['ci'], device=dev)
                launch(g, bp['wt'], gb_up, dtaps, 2 * H, 2 * H, w_tap_stride=bp['ci'] * bp['co'], w_row_stride=bp['co'], w_split=bp['wtc'])
            gbyp = torch.empty_like(x)
            L.check(lib.wgs_upsample2x_bwd(L.ptr(gb_up), L.ptr(gbyp), B, H, H, x.shape[3], st), 'byp_up_bwd')
            g = gx + gbyp
        dz = torch.empty_like(z)
        n = g.numel() // B
        L.check(lib.wgs_linear_dgrad(L.ptr(g), L.ptr(P['lin_w']), None, L.ptr(dz), B, n, z.shape[1], n, z.shape[1],
                                     L.c_float(1.0), L.c_float(1.0), L.c_float(1.0), 0, st), 'seed_linear_dgrad')
        return dz

    def resolve_precision(self, requested=None):
        return C.resolve(self.precision if requested is None else requested, 'sngan', 0)

    def forward(self, z, precision=None):
        return _SN.apply(self, z, self.resolve_precision(precision))


class GenWrapper(nn.Module):
    """sn_gen_resnet.py:57-78 (holds `.model` and the prior's dimensionality)."""

    def __init__(self, model, out_img_shape, dim):
        super().__init__()
        self.model = model
        self.out_img_shape = out_img_shape
        self.distribution = type('D', (), {'dim': dim})()

    def state_dict(self, *a, **k):
        return {'model.' + key: v for key, v in self.model.state_dict().items()}

    def load_state_dict(self, sd, strict=True):
        return self.model.load_state_dict({k[len('model.'):]: v for k, v in sd.items() if k.startswith('model.')}, strict=strict)


def make_resnet_generator(resnet_gen_config, img_size=128, channels=3, latent_dim=128):
    return GenWrapper(GenModel(resnet_gen_config, channels, latent_dim), [channels, img_size, img_size], latent_dim)


class SNGANWrapper(nn.Module):
    """models/gan_load.py:21-28."""

    def __init__(self, G):
        super().__init__()
        self.G = G.model
        self.dim_z = G.distribution.dim

    def forward(self, z, shift=None, precision=None):
        return self.G(z if shift is None else z + shift, precision=precision)

    def resolve_precision(self, requested=None):
        return self.G.resolve_precision(requested)


def build_sngan(pretrained_gan_weights=None, gan_type='SNGAN_MNIST'):
    """models/gan_load.py:31-57 (weights: GenWrapper state_dict, strict=False)."""
    cfg = SNGAN_CONFIG[gan_type]
    G = make_resnet_generator(SN_RES_GEN_CONFIGS[cfg['model']], img_size=cfg['img_size'], channels=cfg['image_channels'],
                              latent_dim=cfg['latent_dim'])
    if pretrained_gan_weights is not None:
        G.load_state_dict(torch.load(pretrained_gan_weights, map_location=torch.device('cpu')), strict=False)
    return SNGANWrapper(G)
