"""BigGAN generator on the HIP kernels — host-side mirror of models/BigGAN/BigGAN.py:54-243,
models/BigGAN/layers.py (SN :58-96, Attention :141-166, ccbn :275-326, bn :330-363, GBlock :372-405) and
BigGANWrapper / build_biggan (models/gan_load.py:65-103).  Parameter / buffer names equal the reference's
state_dict (`shared.weight`, `linear.{weight,bias,u0,sv0}`, `blocks.{i}.0.{conv1,conv2,conv_sc}.*`,
`blocks.{i}.0.bn{1,2}.{gain,bias}.{weight,u0,sv0}`, `blocks.{i}.0.bn{1,2}.stored_{mean,var}`,
`blocks.3.1.{theta,phi,g,o}.*`, `blocks.3.1.gamma`, `output_layer.0.*`, `output_layer.2.*`).

Eval-mode schedule (the generator is frozen on this path):
  * spectral norm: sigma of every SN layer comes from ONE power-iteration step on the stored `u` (layers.py:24-47,
    84-96; `u` is not updated in eval), so W/sigma is a constant — sigma is computed once per device and folded
    into the GEMM epilogue (`alpha`);
  * ccbn / bn: per-sample per-channel affine (+ReLU) kernel; nearest up-sampling folded into the conv gather;
    1x1 shortcut + residual add in the epilogue; tanh in the last epilogue;
  * self-attention at 64x64: 1x1 convs on the same conv family, per-sample score / value products as GEMMs whose
    "weight" operand is the other activation, row-softmax kernel.
Backward propagates only d image -> d z (all six hierarchical z chunks).
"""
import json
import os

import numpy as np
import torch
from torch import nn

from . import _lib as L
from . import conv as C


def generator_arch(ch=64, attention='64'):
    """Channel plan of models/BigGAN/BigGAN.py:13-51 (resolutions 32..512)."""
    att = [int(a) for a in attention.split('_')]
    mk = lambda i, o, res: {'in_channels': [ch * k for k in i], 'out_channels': [ch * k for k in o], 'upsample': [True] * len(i),
                            'resolution': res, 'attention': {r: (r in att) for r in res}}
    return {512: mk([16, 16, 8, 8, 4, 2, 1], [16, 8, 8, 4, 2, 1, 1], [8, 16, 32, 64, 128, 256, 512]),
            256: mk([16, 16, 8, 8, 4, 2], [16, 8, 8, 4, 2, 1], [8, 16, 32, 64, 128, 256]),
            128: mk([16, 16, 8, 4, 2], [16, 8, 4, 2, 1], [8, 16, 32, 64, 128]),
            64: mk([16, 16, 8, 4], [16, 8, 4, 2], [8, 16, 32, 64]),
            32: mk([4, 4, 4], [4, 4, 4], [8, 16, 32])}


# 'mixed' arithmetic of the generator's forward convs: blocks whose OUTPUT resolution is >= the entry run fp16 x2 (2 MFMAs per product), the
# others and the image conv split-bf16 x3; measured per resolution (tests/test_precision_schemes_gpu.py), None = no fp16 block passes the gate.
# The input-gradient convs have no magnitude bound for their fp16 operand and run split-bf16 in every 16-bit mode (conv._desc).
MIXED_FROM_RES = {}


class _SNMixin:
    def _sn_init(self, num_outputs):
        self.register_buffer('u0', torch.randn(1, num_outputs))
        self.register_buffer('sv0', torch.ones(1))

    def sigma(self, eps):
        """One power-iteration step on the stored u (layers.py:24-47 with update=False)."""
        W = self.weight.reshape(self.weight.shape[0], -1)
        v = torch.nn.functional.normalize(self.u0 @ W, eps=eps)
        u = torch.nn.functional.normalize(v @ W.t(), eps=eps)
        return float(((v @ W.t()) @ u.t()).squeeze())


class SNConv2d(nn.Conv2d, _SNMixin):
    def __init__(self, ci, co, kernel_size=3, padding=1, bias=True):
        nn.Conv2d.__init__(self, ci, co, kernel_size, 1, padding, bias=bias)
        self._sn_init(co)


class SNLinear(nn.Linear, _SNMixin):
    def __init__(self, i, o, bias=True):
        nn.Linear.__init__(self, i, o, bias=bias)
        self._sn_init(o)


class ccbn(nn.Module):
    def __init__(self, output_size, input_size):
        super().__init__()
        self.gain = SNLinear(input_size, output_size, bias=False)
        self.bias = SNLinear(input_size, output_size, bias=False)
        self.register_buffer('stored_mean', torch.zeros(output_size))
        self.register_buffer('stored_var', torch.ones(output_size))


class bn(nn.Module):
    def __init__(self, output_size):
        super().__init__()
        self.gain = nn.Parameter(torch.ones(output_size))
        self.bias = nn.Parameter(torch.zeros(output_size))
        self.register_buffer('stored_mean', torch.zeros(output_size))
        self.register_buffer('stored_var', torch.ones(output_size))


class Attention(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.ch = ch
        self.theta = SNConv2d(ch, ch // 8, kernel_size=1, padding=0, bias=False)
        self.phi = SNConv2d(ch, ch // 8, kernel_size=1, padding=0, bias=False)
        self.g = SNConv2d(ch, ch // 2, kernel_size=1, padding=0, bias=False)
        self.o = SNConv2d(ch // 2, ch, kernel_size=1, padding=0, bias=False)
        self.gamma = nn.Parameter(torch.tensor(0.))


class GBlock(nn.Module):
    def __init__(self, ci, co, bn_in):
        super().__init__()
        self.in_channels, self.out_channels = ci, co
        self.conv1 = SNConv2d(ci, co)
        self.conv2 = SNConv2d(co, co)
        self.conv_sc = SNConv2d(ci, co, kernel_size=1, padding=0)
        self.bn1 = ccbn(ci, bn_in)
        self.bn2 = ccbn(co, bn_in)


class _BG(torch.autograd.Function):
    @staticmethod
    def forward(ctx, G, z, y, prec):
        ctx.prec = prec
        img, saved = G._fwd(z, y, ctx.needs_input_grad[1], prec)
        ctx.G, ctx.saved = G, saved
        ctx.hooks = None
        if ctx.needs_input_grad[1]:              # bound to THIS forward's autograd node (stylegan2._Synthesis)
            ctx.hooks, G.bwd_hooks = G.bwd_hooks, None
        if G.debug_keep is not None and saved is not None:     # ReLU gates in execution order, NCHW (tests)
            gates = []
            for (_, _, a1, _, _, a2, _, _) in saved[0]:
                gates += [(a1 > 0).permute(0, 3, 1, 2), (a2 > 0).permute(0, 3, 1, 2)]
            G.debug_keep['gates'] = gates + [(saved[2] > 0).permute(0, 3, 1, 2)]
        return img

    @staticmethod
    def backward(ctx, gimg):
        hooks, ctx.hooks = ctx.hooks, None
        return None, ctx.G._bwd(ctx.saved, gimg.contiguous(), ctx.prec, hooks), None, None


class Generator(nn.Module):
    """Generator(G_ch, dim_z, bottom_width, resolution, G_attn, n_classes, shared_dim, hier, BN_eps, SN_eps, ...) —
    keyword-compatible with the reference constructor for the options its generator_config.json sets
    (G_shared=True, hier=True, G_param='SN', norm_style='bn')."""

    def __init__(self, G_ch=64, dim_z=128, bottom_width=4, resolution=128, G_attn='64', n_classes=1000, shared_dim=0,
                 hier=False, BN_eps=1e-5, SN_eps=1e-12, G_shared=True, G_param='SN', norm_style='bn', **kwargs):
        super().__init__()
        if not (G_shared and hier and G_param == 'SN' and norm_style == 'bn'):
            raise NotImplementedError("HIP BigGAN path implements the reference configuration: G_shared, hier, SN, bn")
        self.ch, self.bottom_width, self.resolution = G_ch, bottom_width, resolution
        self.n_classes, self.BN_eps, self.SN_eps = n_classes, BN_eps, SN_eps
        self.shared_dim = shared_dim if shared_dim > 0 else dim_z
        self.arch = generator_arch(G_ch, G_attn)[resolution]
        self.num_slots = len(self.arch['in_channels']) + 1
        self.z_chunk_size = dim_z // self.num_slots
        self.dim_z = self.z_chunk_size * self.num_slots
        self.shared = nn.Embedding(n_classes, self.shared_dim)
        self.linear = SNLinear(self.dim_z // self.num_slots, self.arch['in_channels'][0] * bottom_width ** 2)
        bn_in = self.shared_dim + self.z_chunk_size
        blocks = []
        for i in range(len(self.arch['out_channels'])):
            blk = [GBlock(self.arch['in_channels'][i], self.arch['out_channels'][i], bn_in)]
            if self.arch['attention'][self.arch['resolution'][i]]:
                blk.append(Attention(self.arch['out_channels'][i]))
            blocks.append(nn.ModuleList(blk))
        self.blocks = nn.ModuleList(blocks)
        self.output_layer = nn.Sequential(bn(self.arch['out_channels'][-1]), nn.ReLU(), SNConv2d(self.arch['out_channels'][-1], 3))
        for p in self.parameters():
            p.requires_grad_(False)
        self._prep = None
        self.bwd_hooks = None        # hook list for the next differentiable forward's backward (trainer.TrainStep, as stylegan2.Generator.bwd_hooks)
        self.debug_keep = None
        self.precision = 'fp32'      # arithmetic of the convs when forward() is not told otherwise (conv.PRECISION_NAMES)
        self.mixed_from_res = MIXED_FROM_RES.get(resolution)      # 'mixed' (conv.AUTO_TABLE): fp16 x2 from this block resolution up, split-bf16 below

    def _apply(self, fn, *a, **k):
        self._prep = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._prep = None
        return super().load_state_dict(*a, **k)

    # -- constants derived from the frozen weights ------------------------------------------------------
    def _prepare(self):
        dev = self.shared.weight.device
        if self._prep is not None and self._prep['dev'] == dev:
            return self._prep
        if dev.type != 'cuda':
            raise L.WgsError("BigGAN Generator runs on the HIP kernels only: move it to the GPU (no CPU fallback)")
        eps = self.SN_eps
        with torch.no_grad():
            def conv_pack(m, pad_co=None):
                co, ci, k, _ = m.weight.shape
                wp = C.pack_weight(m.weight.float())
                b = m.bias.detach().clone() if m.bias is not None else None
                if pad_co:
                    wpp = torch.zeros(pad_co, k * k, ci, device=dev)
                    wpp[:co] = wp
                    bb = torch.zeros(pad_co, device=dev)
                    bb[:co] = b
                    wp, b, co = wpp, bb, pad_co
                wt = C.repack_w_t(wp, co, k * k, ci)
                # frozen weights: the Winograd U operands of 'fp32w' launches are built once per layout, not per call
                return dict(wp=wp, wt=wt, wc=C.WinoCache(wp), wtc=C.WinoCache(wt), b=b, ci=ci, co=co, k=k, inv=1.0 / m.sigma(eps))

            def cc(m):
                r = torch.rsqrt(m.stored_var + self.BN_eps)
                return dict(wg=m.gain.weight.contiguous(), ig=1.0 / m.gain.sigma(eps), wb=m.bias.weight.contiguous(),
                            ib=1.0 / m.bias.sigma(eps), r=r, mean=m.stored_mean, C=m.stored_mean.shape[0])
            s2 = self.bottom_width ** 2
            c0 = self.arch['in_channels'][0]
            perm = torch.arange(c0 * s2, device=dev).reshape(c0, s2).t().reshape(-1)     # (c,h,w) rows -> (h,w,c)
            P = {'dev': dev, 'lin_w': self.linear.weight[perm].contiguous(), 'lin_b': self.linear.bias[perm].contiguous(),
                 'lin_inv': 1.0 / self.linear.sigma(eps), 'c0': c0, 'blocks': []}
            for blk in self.blocks:
                gb = blk[0]
                d = dict(c1=conv_pack(gb.conv1), c2=conv_pack(gb.conv2), sc=conv_pack(gb.conv_sc), bn1=cc(gb.bn1), bn2=cc(gb.bn2),
                         att=None)
                if len(blk) > 1:
                    a = blk[1]
                    d['att'] = dict(theta=conv_pack(a.theta), phi=conv_pack(a.phi), g=conv_pack(a.g), o=conv_pack(a.o),
                                    gamma=float(a.gamma), ch=a.ch)
                P['blocks'].append(d)
            ob = self.output_layer[0]
            r = torch.rsqrt(ob.stored_var + self.BN_eps)
            P['out_scale'] = (ob.gain * r).contiguous()
            P['out_shift'] = (ob.bias - ob.stored_mean * ob.gain * r).contiguous()
            P['out'] = conv_pack(self.output_layer[2], pad_co=8)
        self._prep = P
        return P

    # -- small helpers ------------------------------------------------------------------------------------
    def _pad_k(self, w):
        """Frozen weights whose K is not a multiple of 4 (the 256 / 512 architectures: z chunks of 17 / 15), zero-padded once per
        _prepare() — the cache lives in the per-instance `_prep` dict, which load_state_dict / .to() rebuild."""
        K = w.shape[1]
        Kp = (K + 3) & ~3
        cache = self._prepare().setdefault('padded', {})
        key = (w.data_ptr(), tuple(w.shape))
        if key not in cache:
            wp = torch.zeros(w.shape[0], Kp, device=w.device)
            wp[:, :K] = w
            cache[key] = (wp, w)          # keeps `w` alive: its address cannot be reused while the entry exists
        return cache[key][0], Kp

    def _lin(self, x, w, inv, bias=None, bscale=1.0):
        B, K = x.shape
        if K % 4:
            w, Kp = self._pad_k(w)
            x = torch.nn.functional.pad(x, (0, Kp - K))
            K = Kp
        N = w.shape[0]
        y = torch.empty(B, N, device=x.device)
        L.check(L.lib().wgs_linear_fwd(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), B, N, K, K, N, L.c_float(inv), L.c_float(bscale),
                                       0, 0, L.c_float(0.0), L.c_float(1.0), L.stream()), 'biggan_linear')
        return y

    def _lin_dgrad(self, g, w, inv, out, accumulate):
        B, N = g.shape
        K = w.shape[1]
        if K % 4:
            wp, Kp = self._pad_k(w)
            tmp = torch.empty(B, Kp, device=g.device)
            self._lin_dgrad(g, wp, inv, tmp, accumulate=False)
            if accumulate:
                out += tmp[:, :K]
            else:
                out.copy_(tmp[:, :K])
            return
        L.check(L.lib().wgs_linear_dgrad(L.ptr(g), L.ptr(w), None, L.ptr(out), B, N, K, N, K, L.c_float(inv), L.c_float(1.0),
                                         L.c_float(1.0), int(accumulate), L.stream()), 'biggan_linear_dgrad')

    def _ccbn_affine(self, c, yb):
        """scale/shift [B,C] of ccbn in eval mode: ((x-mean)*rsqrt(var+eps))*(1+gain(y)) + bias(y)."""
        gain = self._lin(yb, c['wg'], c['ig'])
        bias = self._lin(yb, c['wb'], c['ib'])
        scale = ((1.0 + gain) * c['r']).contiguous()
        shift = (bias - c['mean'] * scale).contiguous()
        return scale, shift

    @staticmethod
    def _affine_relu(x, scale, shift):
        B, H, W, Cn = x.shape
        y = torch.empty_like(x)
        L.check(L.lib().wgs_affine_relu_fwd(L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(y), B, H * W, Cn, 1, L.stream()), 'affine_relu')
        return y

    @staticmethod
    def _affine_relu_bwd(x, y, g, scale):
        B, H, W, Cn = x.shape
        dx = torch.empty_like(x)
        dscale, dshift = torch.zeros(B, Cn, device=x.device), torch.zeros(B, Cn, device=x.device)
        L.check(L.lib().wgs_affine_relu_bwd(L.ptr(x), L.ptr(y), L.ptr(g), L.ptr(scale), L.ptr(dx), L.ptr(dscale), L.ptr(dshift),
                                            B, H * W, Cn, 1, L.stream()), 'affine_relu_bwd')
        return dx, dscale, dshift

    @staticmethod
    def _conv(x, c, prec, ups=0, addend=None, act=0, out_hw=None):
        B, H = x.shape[0], x.shape[1] << ups
        k = c['k']
        pad = k // 2
        y = torch.empty(B, H, H, c['co'], device=x.device)
        taps = [(ky - pad, kx - pad, ky * k + kx) for ky in range(k) for kx in range(k)]
        C.launch(x, c['wp'], y, taps, H, H, w_tap_stride=c['ci'], w_row_stride=k * k * c['ci'], ups=ups, alpha=c['inv'], bias=c['b'],
                 addend=addend, act=act, precision=prec, w_split=c.get('wc'))
        return y

    @staticmethod
    def _conv_dgrad(g, c, prec):
        """d/d(input of the conv on the up-sampled grid): [B,H,H,ci].  (fp16 modes: a gradient operand without a magnitude
        bound runs in split-bf16: grad_operand=True)"""
        B, H = g.shape[0], g.shape[1]
        k = c['k']
        pad = k // 2
        dx = torch.empty(B, H, H, c['ci'], device=g.device)
        taps = [(pad - ky, pad - kx, ky * k + kx) for ky in range(k) for kx in range(k)]
        C.launch(g, c['wt'], dx, taps, H, H, w_tap_stride=c['ci'] * c['co'], w_row_stride=c['co'], alpha=c['inv'], precision=prec, grad_operand=True,
                 w_split=c.get('wtc'))
        return dx

    @staticmethod
    def _up_bwd(g):
        B, H2, _, Cn = g.shape
        dx = torch.empty(B, H2 // 2, H2 // 2, Cn, device=g.device)
        L.check(L.lib().wgs_upsample2x_bwd(L.ptr(g), L.ptr(dx), B, H2 // 2, H2 // 2, Cn, L.stream()), 'up_bwd')
        return dx

    # -- self-attention (layers.py:153-166) ---------------------------------------------------------------
    def _att_fwd(self, a, x, save, prec):
        lib, st = L.lib(), L.stream()
        B, H, _, ch = x.shape
        Pq, Pk = H * H, H * H // 4
        theta = self._conv(x, a['theta'], prec)                                     # [B,H,H,ch/8]
        phi_f, g_f = self._conv(x, a['phi'], prec), self._conv(x, a['g'], prec)

        def pool(t):
            Cn = t.shape[3]
            y = torch.empty(B, H // 2, H // 2, Cn, device=x.device)
            idx = torch.empty(B, H // 2, H // 2, Cn, dtype=torch.uint8, device=x.device)
            L.check(lib.wgs_maxpool_fwd(L.ptr(t), L.ptr(y), L.rawptr(idx), B, H, H, Cn, 2, 2, 0, st), 'att_pool')
            return y, idx
        phi, iphi = pool(phi_f)
        g, ig = pool(g_f)
        c8, c2 = ch // 8, ch // 2
        lse = None
        if lib.wgs_attn_supported(B, Pq, Pk, c8, c2):
            # scores, softmax and the weighted sum of g in ONE launch over the batch (csrc/attention.hip): the [B, Pq, Pk] score /
            # attention tensors are never written; the backward recomputes its blocks from the row log-sum-exp `lse`
            o_pre = torch.empty(B, H, H, c2, device=x.device)
            lse = torch.empty(B, Pq, device=x.device)
            L.check(lib.wgs_attn_fwd(L.ptr(theta), L.ptr(phi), L.ptr(g), L.ptr(o_pre), L.ptr(lse), B, Pq, Pk, c8, c2, st), 'attn_fwd')
            beta = None
        else:
            beta, o_pre = self._att_core_unfused(theta, phi, g, B, H, Pq, Pk, c8, c2, prec)
        o = a['o']
        y = torch.empty_like(x)
        C.launch(o_pre, o['wp'], y, [(0, 0, 0)], H, H, w_tap_stride=o['ci'], w_row_stride=o['ci'], alpha=o['inv'] * a['gamma'], addend=x, precision=prec)
        return y, ((x, theta, phi, iphi, g, ig, beta, o_pre, lse) if save else None)

    def _att_core_unfused(self, theta, phi, g, B, H, Pq, Pk, c8, c2, prec):
        """Fallback for attention shapes the fused kernel does not cover: per-sample GEMM -> row softmax -> GEMM (materialises
        the [B, Pq, Pk] score tensor)."""
        lib, st = L.lib(), L.stream()
        scores = torch.empty(B, Pq, Pk, device=theta.device)
        for b in range(B):      # scores[b] = theta[b] (Pq x c8) . phi[b]^T : phi[b] plays the weight operand [Pk, 1, c8]
            C.launch(theta[b].reshape(1, Pq, 1, c8), phi[b], scores[b].reshape(1, Pq, 1, Pk), [(0, 0, 0)], Pq, 1,
                     w_tap_stride=c8, w_row_stride=c8, precision=prec)
        beta = torch.empty_like(scores)
        L.check(lib.wgs_softmax_rows_fwd(L.ptr(scores), L.ptr(beta), L.c_int64(B * Pq), Pk, st), 'att_softmax')
        del scores
        o_pre = torch.empty(B, H, H, c2, device=theta.device)
        for b in range(B):      # o_pre[b] = beta[b] (Pq x Pk) . g[b] (Pk x c2): weight operand = g[b]^T [c2, 1, Pk]
            gt = C.repack_w_t(g[b].reshape(Pk, 1, c2), Pk, 1, c2)            # [1, c2, Pk]
            C.launch(beta[b].reshape(1, Pq, 1, Pk), gt, o_pre[b].reshape(1, Pq, 1, c2), [(0, 0, 0)], Pq, 1, w_tap_stride=Pk * c2,
                     w_row_stride=Pk, precision=prec)
        return beta, o_pre

    def _att_core_unfused_bwd(self, theta, phi, g, beta, do_pre, B, Pq, Pk, c8, c2, prec):
        lib, st = L.lib(), L.stream()
        dbeta = torch.empty_like(beta)
        dg = torch.zeros_like(g)
        for b in range(B):
            # dbeta[b] = do_pre[b] (Pq x c2) . g[b]^T ;  dg[b][k,c] = sum_q beta[b][q,k] do_pre[b][q,c]  (wgrad form)
            C.launch(do_pre[b].reshape(1, Pq, 1, c2), g[b], dbeta[b].reshape(1, Pq, 1, Pk), [(0, 0, 0)], Pq, 1, w_tap_stride=c2,
                     w_row_stride=c2, precision=prec, grad_operand=True)
            C.conv2d_wgrad(do_pre[b].reshape(1, Pq, 1, c2), beta[b].reshape(1, Pq, 1, Pk), dg[b].reshape(Pk, 1, c2), 1)
        ds = torch.empty_like(beta)
        L.check(lib.wgs_softmax_rows_bwd(L.ptr(beta), L.ptr(dbeta), L.ptr(ds), L.c_int64(B * Pq), Pk, st), 'att_softmax_bwd')
        del dbeta
        dtheta = torch.empty_like(theta)
        dphi = torch.zeros_like(phi)
        for b in range(B):
            pt = C.repack_w_t(phi[b].reshape(Pk, 1, c8), Pk, 1, c8)          # [1, c8, Pk]
            C.launch(ds[b].reshape(1, Pq, 1, Pk), pt, dtheta[b].reshape(1, Pq, 1, c8), [(0, 0, 0)], Pq, 1, w_tap_stride=Pk * c8,
                     w_row_stride=Pk, precision=prec, grad_operand=True)
            C.conv2d_wgrad(theta[b].reshape(1, Pq, 1, c8), ds[b].reshape(1, Pq, 1, Pk), dphi[b].reshape(Pk, 1, c8), 1)
        return dtheta, dphi, dg

    def _att_bwd(self, a, sv, gy, prec):
        lib, st = L.lib(), L.stream()
        x, theta, phi, iphi, g, ig, beta, o_pre, lse = sv
        B, H, _, ch = x.shape
        Pq, Pk, c8, c2 = H * H, H * H // 4, ch // 8, ch // 2
        o = a['o']
        do_pre = torch.empty_like(o_pre)
        C.launch(gy, o['wt'], do_pre, [(0, 0, 0)], H, H, w_tap_stride=o['ci'] * o['co'], w_row_stride=o['co'], alpha=o['inv'] * a['gamma'], precision=prec, grad_operand=True)
        if lse is not None:
            dtheta, dphi, dg = torch.empty_like(theta), torch.empty_like(phi), torch.empty_like(g)
            ws = torch.empty(B * Pq, device=x.device)
            L.check(lib.wgs_attn_bwd(L.ptr(theta), L.ptr(phi), L.ptr(g), L.ptr(o_pre), L.ptr(lse), L.ptr(do_pre), L.ptr(ws), L.ptr(dtheta),
                                     L.ptr(dphi), L.ptr(dg), B, Pq, Pk, c8, c2, st), 'attn_bwd')
        else:
            dtheta, dphi, dg = self._att_core_unfused_bwd(theta, phi, g, beta, do_pre, B, Pq, Pk, c8, c2, prec)

        def unpool(d, idx):
            Cn = d.shape[3]
            dx = torch.empty(B, H, H, Cn, device=x.device)
            L.check(lib.wgs_maxpool_bwd(L.ptr(d), L.rawptr(idx), L.ptr(dx), B, H, H, Cn, 2, 2, 0, st), 'att_pool_bwd')
            return dx
        gx = gy + self._conv_dgrad(dtheta, a['theta'], prec)
        gx = gx + self._conv_dgrad(unpool(dphi, iphi), a['phi'], prec)
        gx = gx + self._conv_dgrad(unpool(dg, ig), a['g'], prec)
        return gx

    # -- forward / backward ----------------------------------------------------------------------------------
    def _fwd(self, z, y, save, prec):
        g = self._fwd_gen(z, y, save, prec, None)
        try:
            next(g)
        except StopIteration as e:
            return e.value
        raise L.WgsError("BigGAN pass paused without a pause resolution")

    def _fwd_gen(self, z, y, save, prec, pause_res):
        """The pass as a Python generator (as stylegan2.Generator._synthesis_gen): yields before the first block whose output exceeds
        `pause_res` (an int or a tuple of ascending resolutions, one pause each; at least once when one is given); returns (image, saved)."""
        P = self._prepare()
        pauses = [] if pause_res is None else sorted(pause_res if isinstance(pause_res, (tuple, list)) else [pause_res])
        paused = False
        z, y = z.contiguous(), y.contiguous()
        B = z.shape[0]
        cs = self.z_chunk_size
        zs = [z[:, i * cs:(i + 1) * cs].contiguous() for i in range(self.num_slots)]
        ys = [torch.cat([y, zc], 1).contiguous() for zc in zs[1:]]                       # BigGAN.py:225-227
        s, c0 = self.bottom_width, P['c0']
        h = self._lin(zs[0], P['lin_w'], P['lin_inv'], P['lin_b']).reshape(B, s, s, c0)  # rows pre-permuted to NHWC
        saved = []
        mixed_from = self.mixed_from_res      # 'mixed': fp16 x2 in the blocks whose output is >= this resolution, split-bf16 below
        prec_in = prec
        for d, yb in zip(P['blocks'], ys):
            if C.is_mixed(prec_in):
                prec = 3 if (mixed_from is not None and 2 * h.shape[1] >= mixed_from) else 1
            if pauses and 2 * h.shape[1] > pauses[0]:
                while pauses and 2 * h.shape[1] > pauses[0]:
                    pauses.pop(0)
                paused = True
                yield None
            s1, t1 = self._ccbn_affine(d['bn1'], yb)
            a1 = self._affine_relu(h, s1, t1)
            h1 = self._conv(a1, d['c1'], prec, ups=1)
            s2, t2 = self._ccbn_affine(d['bn2'], yb)
            a2 = self._affine_relu(h1, s2, t2)
            sc = self._conv(h, d['sc'], prec, ups=1)
            out = self._conv(a2, d['c2'], prec, addend=sc)
            att_saved = None
            pre_att = out
            if d['att'] is not None:
                out, att_saved = self._att_fwd(d['att'], out, save, prec)
            if save:
                saved.append((h, s1, a1, h1, s2, a2, yb, att_saved))
            del pre_att
            h = out
        so = P['out_scale'].unsqueeze(0).expand(B, -1).contiguous()
        to = P['out_shift'].unsqueeze(0).expand(B, -1).contiguous()
        af = self._affine_relu(h, so, to)
        if C.is_mixed(prec_in):
            prec = 1          # the image conv (3 output channels, tanh) stays fp32-class
        y8 = self._conv(af, P['out'], prec, act=1)
        img = y8[..., :3].permute(0, 3, 1, 2).contiguous()
        if pause_res is not None and not paused:
            yield None
        return img, ((saved, h, af, so, y8, zs, B) if save else None)

    def _bwd(self, saved_all, gimg, prec, hooks=None):
        P = self._prepare()
        lib, st = L.lib(), L.stream()
        saved, h_last, af, so, y8, zs, B = saved_all
        dev = gimg.device
        Hc = gimg.shape[2]
        g8 = torch.zeros(B, Hc, Hc, 8, device=dev)
        g8[..., :3] = gimg.permute(0, 2, 3, 1)
        dpre = torch.empty_like(g8)
        L.check(lib.wgs_bias_act(L.ptr(g8), None, L.ptr(y8), L.ptr(dpre), 9, 1, L.c_float(0.0), L.c_float(1.0), L.c_int64(g8.numel()),
                                 1, 1, st), 'tanh_bwd')
        gaf = self._conv_dgrad(dpre, P['out'], prec)
        g, _, _ = self._affine_relu_bwd(h_last, af, gaf, so)
        cs = self.z_chunk_size
        dz = torch.zeros(B, self.dim_z, device=dev)
        carrier, hooks = hooks, list(hooks or ())      # [(resolution, callable)]: each called once, at the first block of <= resolution
        if carrier is not None:
            del carrier[:]
        for i in range(len(P['blocks']) - 1, -1, -1):
            d = P['blocks'][i]
            h, s1, a1, h1, s2, a2, yb, att_saved = saved[i]
            while hooks and max(q[0] for q in hooks) >= 2 * h.shape[1]:
                q = max(hooks, key=lambda t_: t_[0])
                hooks.remove(q)
                q[1]()
            if d['att'] is not None:
                g = self._att_bwd(d['att'], att_saved, g, prec)
            dyb = torch.zeros(B, yb.shape[1], device=dev)
            # out = conv2(a2) + conv_sc(up(h))
            ga2 = self._conv_dgrad(g, d['c2'], prec)
            gh1, ds2, dt2 = self._affine_relu_bwd(h1, a2, ga2, s2)
            self._ccbn_grad(d['bn2'], ds2, dt2, dyb)
            ga1 = self._up_bwd(self._conv_dgrad(gh1, d['c1'], prec))
            gh, ds1, dt1 = self._affine_relu_bwd(h, a1, ga1, s1)
            self._ccbn_grad(d['bn1'], ds1, dt1, dyb)
            gsc = self._up_bwd(self._conv_dgrad(g, d['sc'], prec))
            g = gh + gsc
            dz[:, (i + 1) * cs:(i + 2) * cs] = dyb[:, self.shared_dim:]              # ys[i] = cat(y, zs[i+1])
        for q in sorted(hooks, key=lambda t_: -t_[0]):
            q[1]()
        dz0 = torch.empty(B, cs, device=dev)
        self._lin_dgrad(g.reshape(B, -1), P['lin_w'], P['lin_inv'], dz0, accumulate=False)
        dz[:, :cs] = dz0
        return dz

    def _ccbn_grad(self, c, dscale, dshift, dyb):
        """scale = (1+gain)*r, shift = bias - mean*scale  =>  dgain = (dscale - dshift*mean)*r, dbias = dshift."""
        dgain = ((dscale - dshift * c['mean']) * c['r']).contiguous()
        self._lin_dgrad(dgain, c['wg'], c['ig'], dyb, accumulate=True)
        self._lin_dgrad(dshift.contiguous(), c['wb'], c['ib'], dyb, accumulate=True)

    def resolve_precision(self, requested=None):
        return C.resolve(self.precision if requested is None else requested, 'biggan', self.resolution)

    def forward(self, z, y, precision=None):
        """z [B, dim_z], y = self.shared(class ids) [B, shared_dim] (BigGAN.py:222-243).  precision (extension): arithmetic of
        this call's convs (conv.PRECISION_NAMES; default self.precision)."""
        return _BG.apply(self, z, y, self.resolve_precision(precision))


class BigGANWrapper(nn.Module):
    """models/gan_load.py:65-81 (classes drawn with numpy like the reference; tensors follow z's device)."""

    def __init__(self, G, target_classes=(239,)):
        super().__init__()
        self.G = G
        self.target_classes = nn.Parameter(data=torch.tensor(target_classes, dtype=torch.int64), requires_grad=False)
        self.dim_z = self.G.dim_z

    def mixed_classes(self, batch_size):
        if len(self.target_classes.data.shape) == 0:
            return self.target_classes.repeat(batch_size)
        return torch.from_numpy(np.random.choice(self.target_classes.cpu().numpy(), [batch_size]))

    def forward(self, z, shift=None, precision=None, classes=None):
        """classes (extension): class ids [B] of this call; None draws them as the reference does (gan_load.py:73-79)."""
        target_classes = (self.mixed_classes(z.shape[0]) if classes is None else classes).to(z.device)
        return self.G(z if shift is None else z + shift, self.G.shared(target_classes), precision=precision)

    def resolve_precision(self, requested=None):
        return self.G.resolve_precision(requested)

    # -- the un-shifted pass G(z) in stages (extension; trainer.TrainStep, as StyleGAN2Wrapper) ------------------------------
    def begin(self, z, precision=None, pause_res=32, classes=None):
        target_classes = (self.mixed_classes(z.shape[0]) if classes is None else classes).to(z.device)
        return L.StagedPass(self.G._fwd_gen(z, self.G.shared(target_classes), False, self.G.resolve_precision(precision), pause_res))

    @staticmethod
    def advance(handle):
        return handle.advance()

    @staticmethod
    def finish(handle):
        return handle.finish()


def build_biggan(pretrained_gan_weights=None, target_classes=(239,), config_file=None):
    """models/gan_load.py:84-103: generator_config.json -> Generator(**config); weights strict."""
    cfg = dict(G_ch=96, dim_z=120, shared_dim=128, hier=True, G_attn='64', BN_eps=1e-5, SN_eps=1e-6, resolution=128, n_classes=1000)
    if config_file is not None and os.path.isfile(config_file):
        with open(config_file) as f:
            js = json.load(f)
        cfg.update({k: js[k] for k in ('G_ch', 'dim_z', 'shared_dim', 'hier', 'G_attn', 'BN_eps', 'SN_eps') if k in js})
    G = Generator(**cfg)
    if pretrained_gan_weights is not None:
        G.load_state_dict(torch.load(pretrained_gan_weights, map_location=torch.device('cpu')), strict=True)
    return BigGANWrapper(G, target_classes if target_classes is not None else (239,))
