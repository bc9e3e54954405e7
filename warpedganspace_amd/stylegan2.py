"""StyleGAN2 generator on the HIP kernels — host-side mirror of models/StyleGAN2/model.py.

The module tree reproduces the reference's parameter / buffer names and shapes exactly (so a
`g_ema` checkpoint loads with `load_state_dict`, models/gan_load.py:186), but `forward` is an explicit
kernel schedule instead of nn.Module calls:

  * activations are NHWC; the modulated conv never materialises per-sample weights
    (model.py:190-199): style scales the A operand while it is staged, demodulation scales the
    accumulators, noise + bias + leaky-relu*sqrt(2) run in the GEMM epilogue (csrc/conv_igemm.hip);
  * the stride-2 transposed conv (model.py:201-212) is 4 sub-pixel phase GEMMs followed by one fused
    blur + noise + bias + activation pass;
  * all 20 modulation EqualLinears are one [B,512] x [512, sum(Cin)] product;
  * backward is hand-derived and propagates ONLY the input gradient (G is frozen, lib/trainer.py
    never uses its weight gradients): d image -> d latent -> (Z space) d z.
"""
import ctypes
import math

import torch
from torch import nn

from . import _lib as L
from . import conv as C
from . import ops


class StyleGradBatch(ctypes.Structure):
    """ctypes mirror of wgs_style_grad_batch (include/wgs.h)."""
    _fields_ = [('n', ctypes.c_int32), ('B', ctypes.c_int32), ('ld_s', ctypes.c_int32), ('ld_out', ctypes.c_int32)] + \
               [(k, ctypes.c_void_p * 24) for k in ('num', 'demod', 's', 'dsdir', 'wsq', 'dstyle')] + \
               [('Co', ctypes.c_int32 * 24), ('Ci', ctypes.c_int32 * 24), ('scale2', ctypes.c_float * 24)]


class LinearBatch(ctypes.Structure):
    """ctypes mirror of wgs_linear_batch (include/wgs.h)."""
    _fields_ = [('n', ctypes.c_int32), ('M', ctypes.c_int32), ('in_square', ctypes.c_int32), ('epilogue', ctypes.c_int32),
                ('x', ctypes.c_void_p * 16), ('w', ctypes.c_void_p * 16), ('y', ctypes.c_void_p * 16),
                ('N', ctypes.c_int32 * 16), ('K', ctypes.c_int32 * 16), ('ldx', ctypes.c_int32 * 16), ('ldy', ctypes.c_int32 * 16),
                ('wscale', ctypes.c_float * 16), ('eps', ctypes.c_float * 16), ('out_gain', ctypes.c_float * 16)]

SQRT2 = 2 ** 0.5
MAPPING_BWD_FUSED = True      # the mapping network's backward in one launch (tests switch it off to compare with the per-layer launches)


def _dp(t):
    return None if t is None else t.data_ptr()


def make_kernel(k):
    """Normalised outer-product FIR kernel (model.py:18-26)."""
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


# ---- parameter containers (names/shapes = reference state_dict) -----------------------------------
class _Blur(nn.Module):
    def __init__(self, kernel):
        super().__init__()
        self.register_buffer('kernel', kernel)


class _EqualLinear(nn.Module):
    """EqualLinear parameters (model.py:110-136)."""

    def __init__(self, in_dim, out_dim, bias_init=0.0, lr_mul=1.0):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init))
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul


class _ModulatedConv2d(nn.Module):
    """ModulatedConv2d parameters (model.py:148-185)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 blur_kernel=(1, 3, 3, 1)):
        super().__init__()
        self.in_channel, self.out_channel, self.kernel_size = in_channel, out_channel, kernel_size
        self.upsample, self.demodulate = upsample, demodulate
        if upsample:
            self.blur = _Blur(make_kernel(blur_kernel) * 4)     # Blur(..., upsample_factor=2), pad (1,1)
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = _EqualLinear(style_dim, in_channel, bias_init=1)


class _NoiseInjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))


class _Bias(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))


class _ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))


class _StyledConv(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False):
        super().__init__()
        self.conv = _ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample)
        self.noise = _NoiseInjection()
        self.activate = _Bias(out_channel)


class _ToRGB(nn.Module):
    def __init__(self, in_channel, style_dim, upsample=True):
        super().__init__()
        if upsample:
            self.upsample = _Blur(make_kernel((1, 3, 3, 1)) * 4)   # Upsample(factor 2): pad (2,1)
        self.conv = _ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))


class _Noises(nn.Module):
    pass


# ---- autograd glue -----------------------------------------------------------------------------------
class _Mapping(torch.autograd.Function):
    @staticmethod
    def forward(ctx, G, z):
        w, saved = G._mapping_fwd(z, save=ctx.needs_input_grad[1])
        ctx.G, ctx.saved = G, saved
        if G.debug_keep is not None and saved is not None:
            G.debug_keep['mapping'] = [a > 0 for a in saved[1][1:]]
        return w

    @staticmethod
    def backward(ctx, gw):
        return None, ctx.G._mapping_bwd(ctx.saved, gw.contiguous())


class _Synthesis(torch.autograd.Function):
    @staticmethod
    def forward(ctx, G, w, prec, policy=None):
        ctx.prec, ctx.policy = prec, policy                  # the backward runs in the arithmetic (and per-layer table) its forward ran in
        img, saved = G._synthesis_fwd(w, ctx.needs_input_grad[1], prec, policy)
        ctx.G, ctx.saved = G, saved
        # side work for THIS forward's backward (Generator.bwd_hooks): the list object is bound to this autograd node and taken off the
        # generator, so that another forward / backward of the same generator in between neither sees nor consumes it
        ctx.hooks = None
        if ctx.needs_input_grad[1]:
            ctx.hooks, G.bwd_hooks = G.bwd_hooks, None
        if G.debug_keep is not None and saved is not None:   # leaky-relu gates of every StyledConv, NCHW (tests)
            G.debug_keep['synthesis'] = [(o > 0).permute(0, 3, 1, 2) for o in saved[1]]
        return img

    @staticmethod
    def backward(ctx, gimg):
        hooks, ctx.hooks = ctx.hooks, None
        return None, ctx.G._synthesis_bwd(ctx.saved, gimg.contiguous(), ctx.prec, hooks, ctx.policy), None, None


class Generator(nn.Module):
    """Generator(size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1,3,3,1], lr_mlp=0.01)
    — same signature as the reference (model.py:285-333)."""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=(1, 3, 3, 1), lr_mlp=0.01):
        super().__init__()
        if tuple(blur_kernel) != (1, 3, 3, 1):
            raise NotImplementedError("only the reference's [1,3,3,1] blur kernel is implemented")
        self.size, self.style_dim = size, style_dim
        layers = [nn.Identity()]                      # index 0 is PixelNorm in the reference (no params)
        for _ in range(n_mlp):
            layers.append(_EqualLinear(style_dim, style_dim, lr_mul=lr_mlp))
        self.style = nn.Sequential(*layers)
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier,
                         128: 128 * channel_multiplier, 256: 64 * channel_multiplier,
                         512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        self.input = _ConstantInput(self.channels[4])
        self.conv1 = _StyledConv(self.channels[4], self.channels[4], 3, style_dim)
        self.to_rgb1 = _ToRGB(self.channels[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.convs, self.to_rgbs = nn.ModuleList(), nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.noises = _Noises()
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer('noise_{}'.format(layer_idx), torch.randn(1, 1, 2 ** res, 2 ** res))
        in_channel = self.channels[4]
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(_StyledConv(in_channel, out_channel, 3, style_dim, upsample=True))
            self.convs.append(_StyledConv(out_channel, out_channel, 3, style_dim))
            self.to_rgbs.append(_ToRGB(out_channel, style_dim))
            in_channel = out_channel
        self.n_latent = self.log_size * 2 - 2
        for p in self.parameters():       # G is frozen on this path (lib/trainer.py:143-150 puts it in eval)
            p.requires_grad_(False)
        self._prep = None
        self.debug_keep = None   # tests set this to {} to read back the activation gates of a forward
        self._route = {}            # cached kernel-route decisions of this generator's launches (queries of the library, per shape)
        # A LIST set here before a differentiable forward is bound to that forward's autograd node (and taken off the generator); the
        # (resolution, callable) entries it holds when THAT forward's backward runs — they may be appended after the forward — are each
        # called once, when the pass reaches a layer of <= resolution (trainer.TrainStep: side work under the backward's latency-bound tail)
        self.bwd_hooks = None
        # arithmetic of the convs when forward() is not told otherwise (conv.PRECISION_NAMES); the reference's is fp32
        self.precision = 'fp32'
        self.mixed_policy = None  # conv.MixedPolicy override of 'mixed' (None: conv.mixed_policy(size))

    def resolve_precision(self, requested=None):
        """Concrete arithmetic code this generator runs in for `requested` (None: self.precision)."""
        return C.resolve(self.precision if requested is None else requested, 'stylegan2', self.size)

    # -- derived, device-resident packed weights (rebuilt after load_state_dict / .to()) -----------
    def _apply(self, fn, *a, **k):
        self._prep = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._prep = None
        return super().load_state_dict(*a, **k)

    def refresh(self):
        self._prep = None

    def _prepare(self):
        dev = self.input.input.device
        if self._prep is not None and self._prep['dev'] == dev:
            return self._prep
        if dev.type != 'cuda':
            raise L.WgsError("StyleGAN2 Generator runs on the HIP kernels only: move it to the GPU (no CPU fallback)")
        styled = [self.conv1] + list(self.convs)
        rgbs = [self.to_rgb1] + list(self.to_rgbs)
        mods, offs, off = [], [], 0
        P = {'dev': dev, 'layers': [], 'rgbs': []}
        with torch.no_grad():
            for i, sc in enumerate(styled):
                m = sc.conv
                wp = C.pack_weight(m.weight[0].float())             # [Co, 9, Ci]
                wt = C.repack_w_t(wp, m.out_channel, 9, m.in_channel)
                # frozen weights: 16-bit planes (per arithmetic mode, built on first use) for the DMA-fed conv kernels
                wp_s, wt_s = C.SplitCache(wp), C.SplitCache(wt)
                wsq = torch.empty(m.out_channel, m.in_channel, device=dev)
                L.check(L.lib().wgs_sg2_wsq(L.ptr(wp), L.ptr(wsq), m.out_channel, 9, m.in_channel, L.stream()), 'wsq')
                P['layers'].append(dict(
                    wp=wp, wt=wt, wp_s=wp_s, wt_s=wt_s, wsq=wsq, Ci=m.in_channel, Co=m.out_channel, up=m.upsample, scale=m.scale, off=off,
                    noise=getattr(self.noises, 'noise_{}'.format(i)).reshape(-1).contiguous(),
                    noise_w=sc.noise.weight.contiguous(), bias=sc.activate.bias.contiguous(),
                    blur=(m.blur.kernel.contiguous() if m.upsample else None),
                    blur_f=(torch.flip(m.blur.kernel, [0, 1]).contiguous() if m.upsample else None)))
                if m.upsample:
                    # a-priori bound of the layer's output for a forward fp16 plane (wgs_upconv_desc.y_f16): |t| <= sqrt(4 Ci) max|x|
                    # (Cauchy-Schwarz over a phase's <= 4 taps; the demodulation factor is <= 1 / ||w s||), blur gain sum|k|, then
                    # noise, bias and the activation's sqrt(2)
                    ly_ = P['layers'][-1]
                    ly_['plane_mul'] = SQRT2 * float(m.blur.kernel.abs().sum()) * 2.0 * math.sqrt(m.in_channel)
                    ly_['plane_add'] = SQRT2 * (float(sc.noise.weight.abs().max()) * float(ly_['noise'].abs().max()) + float(sc.activate.bias.abs().max()))
                mods.append(m.modulation)
                off += m.in_channel
                # ToRGB sits after conv1 and after every second conv of `convs`
                if i % 2 == 0:
                    r = rgbs[i // 2]
                    P['rgbs'].append(dict(w=r.conv.weight.reshape(3, r.conv.in_channel).contiguous(),
                                          bias=r.bias.reshape(3).contiguous(), scale=r.conv.scale, off=off,
                                          C=r.conv.in_channel,
                                          upk=(r.upsample.kernel.contiguous() if hasattr(r, 'upsample') else None),
                                          upk_f=(torch.flip(r.upsample.kernel, [0, 1]).contiguous()
                                                 if hasattr(r, 'upsample') else None)))
                    mods.append(r.conv.modulation)
                    off += r.conv.in_channel
            P['wmod'] = torch.cat([m.weight for m in mods], 0).contiguous()     # [sumC, style_dim]
            P['bmod'] = torch.cat([m.bias for m in mods], 0).contiguous()
            P['mod_scale'], P['sumC'] = mods[0].scale, off
            P['map'] = [(l.weight.contiguous(), l.bias.contiguous(), l.scale, l.lr_mul) for l in list(self.style)[1:]]
            P['const'] = self.input.input[0].permute(1, 2, 0).contiguous()       # [4,4,C] NHWC
            P['const_amax'] = P['const'].abs().max().reshape(1).contiguous()     # magnitude bound of the first layer's input
            # unit style / identity weight of a fused ToRGB's finish, per number of floats per pixel (4 per 128-channel block of partial sums)
            P['rgb_finish'] = {nc: (torch.ones(1024, nc, device=dev), torch.eye(3, 4, device=dev).repeat(1, nc // 4).contiguous()) for nc in (4, 8, 16)}
        self._prep = P
        return P

    # -- mapping network: PixelNorm + n_mlp x (EqualLinear + fused leaky-relu), model.py:288-295 -------
    def _mapping_fwd(self, z, save):
        P = self._prepare()
        z = z.contiguous()
        B, d = z.shape
        lib, st = L.lib(), L.stream()
        mp = P['map']
        if d == 512 and 1 <= len(mp) <= 16 and all(m[0].shape == (512, 512) and m[2] == mp[0][2] and m[3] == mp[0][3] for m in mp):
            # the whole mapping network in one launch (wgs_mapping_mlp_fwd); acts[l] are views of one buffer
            n = len(mp)
            buf = torch.empty(n + 1, B, d, device=z.device)
            wp = (ctypes.c_void_p * n)(*[m[0].data_ptr() for m in mp])
            bp = (ctypes.c_void_p * n)(*[m[1].data_ptr() for m in mp])
            L.check(lib.wgs_mapping_mlp_fwd(L.ptr(z), wp, bp, L.ptr(buf), B, d, n, L.c_float(mp[0][2]), L.c_float(mp[0][3]),
                                            L.c_float(1e-8), st), 'mapping_mlp_fwd')
            acts = list(buf.unbind(0))
            x = acts[-1]
        else:
            x = torch.empty_like(z)
            L.check(lib.wgs_pixelnorm_fwd(L.ptr(z), L.ptr(x), B, d, L.c_float(1e-8), st), 'pixelnorm')
            acts = [x]
            for (w, b, scale, lr_mul) in mp:
                y = torch.empty(B, w.shape[0], device=z.device)
                L.check(lib.wgs_linear_fwd(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, w.shape[0], w.shape[1], d, w.shape[0],
                                           L.c_float(scale), L.c_float(lr_mul), 0, 1, L.c_float(0.0), L.c_float(1.0), st),
                        'linear_fwd')
                acts.append(y)
                x = y
        return x, ((z, acts) if save else None)

    def _mapping_bwd(self, saved, gw):
        P = self._prepare()
        z, acts = saved
        B, d = z.shape
        lib, st = L.lib(), L.stream()
        g = gw
        mp = P['map']
        fused = (acts[0].data_ptr() + 4 * B * d == acts[1].data_ptr()) if len(acts) > 1 else False       # acts are views of the fused forward's one buffer
        if MAPPING_BWD_FUSED and fused and d == 512 and all(m[0].shape == (512, 512) and m[2] == mp[0][2] for m in mp):
            # all L layers in one launch (wgs_mapping_mlp_bwd): nine 44-us launches at the end of the backward become two
            n = len(mp)
            wp = (ctypes.c_void_p * n)(*[m[0].data_ptr() for m in mp])
            gx = torch.empty(B, d, device=z.device)
            L.check(lib.wgs_mapping_mlp_bwd(L.ptr(g), wp, L.rawptr(acts[0]), L.ptr(gx), B, d, n, L.c_float(mp[0][2]), st), 'mapping_mlp_bwd')
            gz = torch.empty_like(z)
            L.check(lib.wgs_pixelnorm_bwd(L.ptr(z), L.ptr(gx), L.ptr(gz), B, d, L.c_float(1e-8), st), 'pixelnorm_bwd')
            return gz
        for i in range(len(P['map']) - 1, -1, -1):
            w, _, scale, _ = P['map'][i]
            gx = torch.empty(B, w.shape[1], device=z.device)
            L.check(lib.wgs_linear_dgrad(L.ptr(g), L.ptr(w), L.ptr(acts[i + 1]), L.ptr(gx), B, w.shape[0], w.shape[1],
                                         w.shape[0], w.shape[1], L.c_float(scale), L.c_float(0.2), L.c_float(SQRT2), 0, st),
                    'linear_dgrad')
            g = gx
        gz = torch.empty_like(z)
        L.check(lib.wgs_pixelnorm_bwd(L.ptr(z), L.ptr(g), L.ptr(gz), B, d, L.c_float(1e-8), st), 'pixelnorm_bwd')
        return gz

    # -- synthesis network, model.py:389-403 ---------------------------------------------------------------
    def _synthesis_fwd(self, w, save, prec, policy=None):
        g = self._synthesis_gen(w, save, prec, None, policy)
        try:
            next(g)
        except StopIteration as e:
            return e.value
        raise L.WgsError("synthesis generator paused without a pause resolution")

    def synthesis_begin(self, w, prec, pause_res, policy=None):
        """Enqueue the synthesis layers whose output is <= pause_res (nothing saved) and return a handle for synthesis_advance() /
        synthesis_finish().  `pause_res` may be a tuple of ascending resolutions: the pass then pauses before the first layer above
        each.  The low-resolution layers are latency-bound (a tenth of the FLOPs, a quarter of the pass's time): the training step
        runs them for the NEXT batch's un-shifted pass next to the same layers of this batch's shifted pass (trainer.TrainStep)."""
        return L.StagedPass(self._synthesis_gen(w.contiguous(), False, prec, pause_res, policy))

    @staticmethod
    def synthesis_advance(handle):
        """Enqueue the layers up to the next pause resolution: None while the pass is paused again, the image when it is complete."""
        return handle.advance()

    @staticmethod
    def synthesis_finish(handle):
        """Enqueue everything that is left: the image."""
        return handle.finish()

    def _synthesis_gen(self, w, save, prec, pause_res, policy=None):
        """The synthesis pass as a Python generator: yields before the first layer whose output exceeds `pause_res` (an int, or a tuple of
        ascending resolutions with one pause each; None: never; at least once when a pause resolution is given) and returns (image, saved)."""
        P = self._prepare()
        # per-layer table of the 'mixed' modes: this call's (a step engine passes its calibrated one), else the instance's override, else the default
        pol = policy or self.mixed_policy or C.mixed_policy(self.size, prec)
        lib, st = L.lib(), L.stream()
        w = w.contiguous()
        B = w.shape[0]
        dev = w.device
        sumC = P['sumC']
        S = torch.empty(B, sumC, device=dev)      # every layer's modulation output s[b, ci] (bias_init = 1)
        L.check(lib.wgs_linear_fwd(L.ptr(w), L.ptr(P['wmod']), L.ptr(P['bmod']), L.ptr(S), B, sumC, self.style_dim,
                                   self.style_dim, sumC, L.c_float(P['mod_scale']), L.c_float(1.0), 0, 0,
                                   L.c_float(0.0), L.c_float(1.0), st), 'modulation')
        x = P['const'].unsqueeze(0).expand(B, -1, -1, -1).contiguous()
        outs = []
        skip = None
        # fp16 modes: magnitude chain for the dynamic operand scale — every producer raises max|y| of its output (one atomic
        # per wave in its epilogue), the consumer scales its operand x * style by a power of two taken from max|x| * max|style|
        # before rounding it to fp16: a forward pass cannot overflow fp16 whatever the checkpoint's activation magnitudes
        f16_chain = any(C.layer_precision(prec, 4 << ((j + 1) // 2), ly_['up'], pol) in (2, 3) for j, ly_ in enumerate(P['layers']))
        xmax = [P['const_amax']] + list(torch.zeros(len(P['layers']), 1, device=dev).unbind(0)) if f16_chain else None
        smax = S.abs().max().reshape(1) if f16_chain else None
        # every layer's demodulation vector scale * rsqrt(scale^2 * sum_i s^2 wsq + 1e-8) in one batched launch
        demods = [torch.empty(B, ly['Co'], device=dev) for ly in P['layers']]
        nl = len(P['layers'])
        if nl <= 16 and all(ly['Ci'] <= 512 for ly in P['layers']):
            lb = LinearBatch()
            lb.n, lb.M, lb.in_square, lb.epilogue = nl, B, 1, 2
            for i, ly in enumerate(P['layers']):
                lb.x[i] = S.data_ptr() + 4 * ly['off']
                lb.w[i], lb.y[i] = ly['wsq'].data_ptr(), demods[i].data_ptr()
                lb.N[i], lb.K[i], lb.ldx[i], lb.ldy[i] = ly['Co'], ly['Ci'], sumC, ly['Co']
                lb.wscale[i], lb.eps[i], lb.out_gain[i] = ly['scale'] ** 2, 1e-8, ly['scale']
            L.check(lib.wgs_linear_fwd_batch(ctypes.byref(lb), st), 'demod batch')
        else:
            for i, ly in enumerate(P['layers']):
                L.check(lib.wgs_linear_fwd(L.rawptr(S[:, ly['off']:]), L.ptr(ly['wsq']), None, L.ptr(demods[i]), B, ly['Co'], ly['Ci'],
                                           sumC, ly['Co'], L.c_float(ly['scale'] ** 2), L.c_float(0.0), 1, 2, L.c_float(1e-8),
                                           L.c_float(ly['scale']), st), 'demod')
        xplane = None          # (fp16 operand plane, its magnitude bound) of the current layer's input, written by the producing up-conv
        paused = False
        pauses = [] if pause_res is None else sorted(pause_res if isinstance(pause_res, (tuple, list)) else [pause_res])
        for i, ly in enumerate(P['layers']):
            Ci, Co = ly['Ci'], ly['Co']
            s_view = S[:, ly['off']:]
            demod = demods[i]
            H = (x if x is not None else xplane[0]).shape[1]
            if pauses and (2 * H if ly['up'] else H) > pauses[0]:
                while pauses and (2 * H if ly['up'] else H) > pauses[0]:
                    pauses.pop(0)
                paused = True
                yield None
            lp = C.layer_precision(prec, 2 * H if ly['up'] else H, ly['up'], pol)      # 'mixed': per-layer arithmetic
            if lp == C.BF16W and Co % 128:       # the F(2,3) split-bf16 form covers >= 128 output channels: the few-channel layers are plain split-bf16 launches
                lp = 1
            sc_kw = dict(a_amax=xmax[i], a_amax2=smax) if (f16_chain and lp in (2, 3)) else {}
            ymax = xmax[i + 1] if f16_chain else None
            rgbp = None
            if xplane is not None:
                # the producing up-conv wrote this conv's fp16 operand plane (style and scale folded in): staged as it is
                rgb_kw = {}
                if i % 2 == 0 and C.rgb_fused_ok(B, H, Ci, Co, lp, True):
                    # ... and its ToRGB runs in this conv's epilogue; without a backward to feed, the layer's output is not stored at all
                    r_ = P['rgbs'][i // 2]
                    rgbp = torch.empty(B, H, H, 4, device=dev)
                    rgb_kw = dict(rgb=dict(out=rgbp, s=S[:, r_['off']:], ld=sumC, w=r_['w'], scale=r_['scale']))
                keep = save or not rgb_kw or i + 1 < len(P['layers'])
                y = C.conv2d(xplane[0], ly['wp'], 3, pad=1, col_scale=demod, noise=ly['noise'], noise_w=ly['noise_w'], bias=ly['bias'],
                             act_slope=0.2, gain=SQRT2, w_split=ly['wp_s'], precision=lp, y_amax=ymax, a_amax=xplane[1], a_bound=1.0, x_f16=True,
                             out=torch.empty(B, H, H, Co, device=dev) if keep else C.NoOutput(B, H, H, Co), **rgb_kw)
                if not keep:
                    y = None
                xplane = None
            elif ly['up'] and C.upconv_fused_ok(H, Ci, Co, lp):
                nxt = P['layers'][i + 1] if i + 1 < len(P['layers']) else None
                if nxt is not None and f16_chain and C.fwd_plane_ok(B, 2 * H, Co, nxt['Co'], C.layer_precision(prec, 2 * H, False, pol)):
                    y, yh, yb = C.upconv_blur_act(x, ly['wp_s'], ly['blur'], s_view, sumC, demod, ly['noise'], ly['noise_w'], ly['bias'], lp,
                                                  y_amax=ymax, plane=dict(scale=S[:, nxt['off']:], ld=sumC, mul=ly['plane_mul'], add=ly['plane_add'],
                                                                          keep_y=save), **sc_kw)
                    xplane = (yh, yb)
                else:
                    y = C.upconv_blur_act(x, ly['wp_s'], ly['blur'], s_view, sumC, demod, ly['noise'], ly['noise_w'], ly['bias'], lp,
                                          y_amax=ymax, **sc_kw)
            elif ly['up']:
                t = C.conv_transpose2d_s2(x, ly['wp'], a_scale=s_view, a_ld=sumC, col_scale=demod, w_split=ly['wp_s'], precision=lp,
                                          **sc_kw)
                y = torch.empty(B, 2 * H, 2 * H, Co, device=dev)
                L.check(lib.wgs_sg2_blur_noise_bias_act(L.ptr(t), L.ptr(ly['blur']), L.ptr(ly['noise']),
                                                        L.ptr(ly['noise_w']), L.ptr(ly['bias']), L.ptr(y), L.ptr(ymax), B, 2 * H,
                                                        2 * H, Co, st), 'blur_nba')
                del t
            else:
                ckw = dict(a_scale=s_view, a_ld=sumC, col_scale=demod, noise=ly['noise'], noise_w=ly['noise_w'], bias=ly['bias'], act_slope=0.2,
                           gain=SQRT2, w_split=ly['wp_s'], y_amax=ymax, **sc_kw)
                rgb_kw, keep = {}, True
                key = ('rgb_halo', i, B, lp)
                if i % 2 == 0 and lp == C.BF16W and Co in (128, 256, 512):
                    # the F(2,3) split-bf16 kernel holds 128 output channels of its pixels: ToRGB in its epilogue — the whole sum at 128 channels
                    # (StyleGAN2-256's last layer: without a backward to feed, the layer's output is not stored at all), one partial sum per
                    # 128-channel block at 256 / 512 (the finishing launch adds them: the layer's output is not read back by a ToRGB launch)
                    key = ('rgb_w16', i, B)
                    if key not in self._route:
                        self._route[key] = C.rgb_wino16_ok(x, ly['wp'], **ckw)
                    if self._route[key]:
                        r_ = P['rgbs'][i // 2]
                        rgbp = torch.empty(B, H, H, 4 * (Co // 128), device=dev)
                        rgb_kw = dict(rgb=dict(out=rgbp, s=S[:, r_['off']:], ld=sumC, w=r_['w'], scale=r_['scale']))
                        keep = save or i + 1 < len(P['layers'])
                elif i % 2 == 0 and Co <= 64:
                    # StyleGAN2-1024's 64- / 32-channel layers at 512^2 / 1024^2: ToRGB in the few-channel kernel's epilogue (decided once per
                    # (layer, batch, arithmetic)); without a backward to feed, the last layer's output is not stored at all
                    if key not in self._route:
                        self._route[key] = C.rgb_halo_ok(x, ly['wp'], lp, **ckw)
                    if self._route[key]:
                        r_ = P['rgbs'][i // 2]
                        rgbp = torch.empty(B, H, H, 4, device=dev)
                        rgb_kw = dict(rgb=dict(out=rgbp, s=S[:, r_['off']:], ld=sumC, w=r_['w'], scale=r_['scale']))
                        keep = save or i + 1 < len(P['layers'])
                y = C.conv2d(x, ly['wp'], 3, pad=1, precision=lp, out=None if keep else C.NoOutput(B, H, H, Co), **ckw, **rgb_kw)
                if not keep:
                    y = None
            outs.append(y)
            x = y
            if i % 2 == 0:
                r = P['rgbs'][i // 2]
                Hc = H if not ly['up'] else 2 * H
                img = torch.empty(B, 3, Hc, Hc, device=dev)
                if rgbp is not None:
                    # the channel sums came out of the conv's epilogue: bias + up-sampled skip through the same kernel (C = 4, unit style)
                    nc = rgbp.shape[3]              # 4 floats per 128-channel block of the producing conv
                    one4, eye34 = P['rgb_finish'][nc]
                    if B > one4.shape[0]:
                        one4 = torch.ones(B, nc, device=dev)
                    L.check(lib.wgs_sg2_torgb_up_fwd(L.ptr(rgbp), L.ptr(one4[:B]), nc, L.ptr(eye34), L.ptr(r['bias']), L.ptr(skip),
                                                     L.ptr(r['upk']), L.ptr(img), B, Hc, Hc, nc, L.c_float(1.0), st), 'torgb_finish')
                elif skip is not None:    # + Upsample(skip) (model.py:279-281), evaluated inside the ToRGB kernel's epilogue
                    L.check(lib.wgs_sg2_torgb_up_fwd(L.ptr(x), L.rawptr(S[:, r['off']:]), sumC, L.ptr(r['w']), L.ptr(r['bias']), L.ptr(skip),
                                                     L.ptr(r['upk']), L.ptr(img), B, Hc, Hc, r['C'], L.c_float(r['scale']), st), 'torgb_up')
                else:
                    # the plain ToRGB kernel reads its style with row stride C: hand it a compact copy of the slice
                    s_rgb = S[:, r['off']:r['off'] + r['C']].contiguous()
                    L.check(lib.wgs_sg2_torgb_fwd(L.ptr(x), L.ptr(s_rgb), L.ptr(r['w']), L.ptr(r['bias']), None,
                                                  L.ptr(img), B, Hc * Hc, r['C'], L.c_float(r['scale']), st), 'torgb')
                skip = img
        saved = (S, outs, demods, B) if save else None
        if pause_res is not None and not paused:
            yield None                    # (a generator smaller than the pause resolution: everything ran in the first stage)
        return skip, saved

    def _synthesis_bwd(self, saved, dimg, prec, hooks=None, policy=None):
        """d image [B,3,S,S] -> d latent [B, style_dim] (all n_latent copies of w summed)."""
        P = self._prepare()
        pol = policy or self.mixed_policy or C.mixed_policy(self.size, prec)
        lib, st = L.lib(), L.stream()
        S, outs, demods, B = saved
        dev = dimg.device
        sumC = P['sumC']
        layers, rgbs = P['layers'], P['rgbs']
        # every zero-initialised accumulator of the pass (style-gradient partial sums, magnitude scalars) in ONE memset
        nz = B * sumC + 16 + B * layers[0]['Ci'] + sum(3 * B * ly['Co'] for ly in layers) + 4 * (3 * len(layers) + 3)
        zbuf, zoff = torch.zeros(nz, device=dev), [0]

        def zeros(*shape):
            n = 1
            for v in shape:
                n *= v
            t = zbuf[zoff[0]:zoff[0] + n].view(*shape)
            zoff[0] += (n + 3) & ~3                  # 16-byte aligned pieces (the kernels use float4 accesses)
            return t
        dS = zeros(B, sumC)                          # d loss / d modulation outputs, all layers
        dskip = dimg
        amax = zeros(len(layers))                    # per layer: max |dy * demod| (magnitude bound of the fp16 dgrad operand)
        gam = zeros(len(layers))                     # per layer: max |gA| of its gradient conv (16-bit kernels' epilogue)
        gA_amax, dimg_amax, skip_level = None, None, 0
        gA, sA_off = None, None                      # un-scaled dgrad of the consumer conv, its style slice
        sg = []                                      # style-gradient reductions of the pass, launched together at the end
        num_next = None
        carrier, hooks = hooks, list(hooks or ())
        if carrier is not None:
            del carrier[:]                       # consumed: the step sees an empty list when every hook has been taken over by this pass
        for i in range(len(layers) - 1, -1, -1):
            ly = layers[i]
            Co = ly['Co']
            out = outs[i]
            Hc = out.shape[1]
            while hooks and max(h[0] for h in hooks) >= Hc:
                h = max(hooks, key=lambda q: q[0])
                hooks.remove(h)
                h[1]()
            Pn = Hc * Hc
            has_rgb = (i % 2 == 0)
            r = rgbs[i // 2] if has_rgb else None
            num = zeros(B, Co)
            dsA = zeros(B, Co) if gA is not None else None
            dsR = zeros(B, Co) if has_rgb else None
            sA = S[:, sA_off:] if gA is not None else None           # rows of the style matrix S, stride sumC
            sR = S[:, r['off']:] if has_rgb else None
            lp = C.layer_precision_bwd(prec, Hc, ly['up'], pol)
            wino, lp = lp == C.FP32W, (0 if lp == C.FP32W else lp)       # 'fp32w': fp32 throughout, Winograd form of the stride-1 gradient conv
            if lp == C.BF16W and ly['Ci'] % 128:
                lp = 1
            # a stride-1 layer's dy has ONE consumer, its gradient conv: in the plain-fp16 launches that fill the chip it is stored
            # only as that conv's fp16 operand plane, scaled from an a-priori bound of its magnitude (its own maximum is not known
            # before the kernel has run): max|gA| from the producing conv's epilogue, max|drgb| <= 4^levels * max|dimg|
            plane = (not ly['up']) and C.dy_plane_ok(B, Hc, Co, ly['Ci'], lp) and (gA is None or gA_amax is not None)
            # an up-sampling layer's dy has one consumer too, the transposed blur: where that writes the gradient conv's fp16 plane, dy reaches it
            # as an fp16 plane as well (same a-priori bound; conv.DY_PLANE_UP)
            if ly['up'] and C.DY_PLANE_UP and C.blur_bwd_f16_ok(B, Hc, Co, ly['Ci'], lp) and (gA is None or gA_amax is not None):
                plane = True
            rgb_args = (L.ptr(dskip if has_rgb else None), L.ptr(r['w']) if has_rgb else None, L.rawptr(sR),
                        L.c_float(r['scale'] if has_rgb else 0.0))
            if plane:
                if has_rgb and dimg_amax is None:
                    dimg_amax = dimg.abs().amax().reshape(1)
                L.check(lib.wgs_sg2_dy_bound(L.rawptr(gA_amax), L.rawptr(sA), L.rawptr(dimg_amax if has_rgb else None),
                                             L.c_float(4.0 ** skip_level), rgb_args[1], rgb_args[2], rgb_args[3], L.ptr(demods[i]),
                                             L.rawptr(amax[i:]), B, Co, sumC, st), 'sg2_dy_bound')
                dy = torch.empty(out.shape, device=dev, dtype=torch.int16)
                L.check(lib.wgs_sg2_act_bwd_f16(L.ptr(out), L.ptr(gA), L.rawptr(sA), *rgb_args, L.ptr(ly['noise']), L.ptr(ly['noise_w']),
                                                L.ptr(ly['bias']), L.ptr(dy, torch.int16), L.rawptr(amax[i:]), L.ptr(num), L.ptr(dsA),
                                                L.ptr(dsR), L.ptr(demods[i]), None, B, Pn, Co, sumC, st), 'sg2_act_bwd_f16')
            else:
                dy = torch.empty_like(out)
                L.check(lib.wgs_sg2_act_bwd(L.ptr(out), L.ptr(gA), L.rawptr(sA), *rgb_args, L.ptr(ly['noise']), L.ptr(ly['noise_w']),
                                            L.ptr(ly['bias']), L.ptr(dy), L.ptr(num), L.ptr(dsA), L.ptr(dsR), L.ptr(demods[i]),
                                            L.rawptr(amax[i:]), B, Pn, Co, sumC, st),
                        'sg2_act_bwd')           # dy is stored already multiplied by this layer's demodulation vector
            # style gradient of the consumer conv (layer i+1) is now complete: direct term dsA + demod path
            if gA is not None:
                c = layers[i + 1]
                sg.append((num_next, demods[i + 1], S[:, c['off']:], dsA, c['wsq'], dS[:, c['off']:], c['Co'], c['Ci']))
            if has_rgb:
                sg.append((None, None, S[:, r['off']:], dsR, None, dS[:, r['off']:], 3, Co))
                if i > 0:   # gradient of the up-sampled skip w.r.t. the lower-resolution image (upfirdn2d.py:110-115)
                    g = ops.upfirdn2d_mhwc(dskip.reshape(B * 3, Hc, Hc, 1), r['upk_f'], 1, 1, 2, 2, 1, 1, 1, 1)
                    dskip = g.reshape(B, 3, Hc // 2, Hc // 2)
                    skip_level += 1                  # |dskip| <= 4 x the level above: the kernel sums to 4
            # input gradient of this layer (un-scaled by its own style: the producer applies it); its maximum feeds the next bound
            gA_amax = gam[i:]
            if ly['up']:
                # |dt| <= 4 max|dy|: the kernel sums to 4
                if C.blur_bwd_f16_ok(B, Hc, Co, ly['Ci'], lp):     # dt only as the gradient conv's fp16 operand plane
                    dt = C.blur_bwd_f16(dy, ly['blur_f'], amax[i:], 4.0)
                    gA = C.conv_transpose2d_s2_dgrad(dt, ly['wt'], w_split=ly['wt_s'], a_amax=amax[i:], a_bound=4.0, precision=lp, x_f16=True,
                                                     y_amax=gA_amax)
                else:
                    dt = ops.upfirdn2d_mhwc(dy, ly['blur_f'], 1, 1, 1, 1, 2, 2, 2, 2)
                    gA = C.conv_transpose2d_s2_dgrad(dt, ly['wt'], w_split=ly['wt_s'], a_amax=amax[i:], a_bound=4.0, precision=lp,
                                                     y_amax=gA_amax if lp >= 1 else None)
                    if lp < 1:
                        gA_amax = None
                del dt
            else:
                gA = C.conv2d_dgrad(dy, ly['wt'], (Hc, Hc), 3, pad=1, w_split=ly['wt_s'], a_amax=amax[i:], a_bound=1.0, precision=C.FP32W if wino else lp,
                                    x_f16=plane, y_amax=gA_amax if lp >= 1 else None)
                if lp < 1:
                    gA_amax = None
            del dy
            sA_off, num_next = ly['off'], num
        for h in sorted(hooks, key=lambda q: -q[0]):
            h[1]()
        # bottom layer: its input is the ConstantInput -> only the style gradient remains
        ly = layers[0]
        ds0 = zeros(B, ly['Ci'])
        L.check(lib.wgs_xg_reduce(L.ptr(P['const']), 0, L.ptr(gA), L.ptr(ds0), B, 16, ly['Ci'], st), 'xg_reduce')
        sg.append((num_next, demods[0], S[:, ly['off']:], ds0, ly['wsq'], dS[:, ly['off']:], ly['Co'], ly['Ci']))
        # every style gradient of the pass (13 modulated convs + 7 ToRGBs at 256^2) in one launch: dS is only read below
        for k0 in range(0, len(sg), 24):
            part = sg[k0:k0 + 24]
            sb = StyleGradBatch()
            sb.n, sb.B, sb.ld_s, sb.ld_out = len(part), B, sumC, sumC
            for k, (num_, dem_, s_, dsd_, wsq_, out_, co_, ci_) in enumerate(part):
                sb.num[k], sb.demod[k], sb.wsq[k] = _dp(num_), _dp(dem_), _dp(wsq_)
                sb.s[k], sb.dsdir[k], sb.dstyle[k] = s_.data_ptr(), dsd_.data_ptr(), out_.data_ptr()
                sb.Co[k], sb.Ci[k], sb.scale2[k] = co_, ci_, 1.0      # the stored demod carries the conv's weight scale: scale2 = 1
            L.check(lib.wgs_sg2_style_grad_batch(ctypes.byref(sb), st), 'style_grad_batch')
        dw = torch.empty(B, self.style_dim, device=dev)
        L.check(lib.wgs_linear_dgrad(L.ptr(dS), L.ptr(P['wmod']), None, L.ptr(dw), B, sumC, self.style_dim, sumC,
                                     self.style_dim, L.c_float(P['mod_scale']), L.c_float(1.0), L.c_float(1.0), 0, st),
                'dlatent')
        return dw

    # -- reference-facing API -------------------------------------------------------------------------------
    def get_latent(self, input):
        return _Mapping.apply(self, input)

    def forward(self, styles, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, noise=None, randomize_noise=False, precision=None, policy=None):
        """Same call signature as the reference (model.py:359-408).  Supported on the HIP path: a single
        style code, the registered noise buffers, no truncation — i.e. exactly what StyleGAN2Wrapper
        (models/gan_load.py:157-179) issues.  `precision` (extension): arithmetic of this call's convs, default self.precision;
        `policy` (extension): conv.MixedPolicy of this call under a 'mixed' mode (default: self.mixed_policy, else the architecture's table)."""
        if len(styles) != 1 or inject_index is not None or noise is not None or randomize_noise or truncation < 1:
            raise NotImplementedError("HIP StyleGAN2 path supports one style code, registered noise, truncation=1")
        s = styles[0]
        if s.ndim != 2:
            raise NotImplementedError("per-layer (W+) latents are not supported on the HIP path")
        w = s if input_is_latent else _Mapping.apply(self, s)
        img = _Synthesis.apply(self, w, self.resolve_precision(precision), policy)
        if return_latents:
            return img, w.unsqueeze(1).repeat(1, self.n_latent, 1)
        return img, None
