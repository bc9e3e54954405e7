"""ProgGAN generator on the HIP kernels — host-side mirror of models/ProgGAN/model.py:12-95 and
ProgGANWrapper (models/gan_load.py:109-120).  Parameter names/shapes equal the reference's state_dict
(`features.{i}.conv.weight`, `features.{i}.wscale.{scale,b}`, `output.conv.weight`, `output.wscale.*`).

Block = PixelNorm -> (nearest 2x upsample) -> conv (no bias) -> WScale (x*scale + b) -> leaky-relu(0.2):
the upsample is folded into the implicit-GEMM gather (descriptor `ups`), WScale + leaky-relu into its
epilogue (`alpha`, `bias`, `act_slope`); PixelNorm is a row-wise kernel on the NHWC tensor.  Backward
propagates only the input gradient (G is frozen).

`num_blocks` < 18 truncates the network (e.g. 14 blocks = 256x256) for tests and for the 256^2 variant of
BASELINE config 2; the default is the reference's fixed 1024^2 network.
"""
import torch
from torch import nn

from . import _lib as L
from . import conv as C

# (in_channels, out_channels, kernel, padding, upsample) — models/ProgGAN/model.py:68-86
BLOCKS = [(512, 512, 4, 3, False), (512, 512, 3, 1, False), (512, 512, 3, 1, True), (512, 512, 3, 1, False),
          (512, 512, 3, 1, True), (512, 512, 3, 1, False), (512, 512, 3, 1, True), (512, 512, 3, 1, False),
          (512, 256, 3, 1, True), (256, 256, 3, 1, False), (256, 128, 3, 1, True), (128, 128, 3, 1, False),
          (128, 64, 3, 1, True), (64, 64, 3, 1, False), (64, 32, 3, 1, True), (32, 32, 3, 1, False),
          (32, 16, 3, 1, True), (16, 16, 3, 1, False)]


PN_FUSED = True        # PixelNorm of the few-channel layers' operand inside the conv kernel (False: the separate pass everywhere)
F16_GRADS = True       # fp16 gradient convs in the fp16 modes (False: split-bf16, as before the magnitude chain existed)


class _WScale(nn.Module):
    def __init__(self, size):
        super().__init__()
        self.scale = nn.Parameter(torch.randn([1]))
        self.b = nn.Parameter(torch.randn(size))


class _Block(nn.Module):
    def __init__(self, ci, co, k, pad):
        super().__init__()
        self.conv = nn.Conv2d(ci, co, k, 1, pad, bias=False)
        self.wscale = _WScale(co)


class _PG(torch.autograd.Function):
    @staticmethod
    def forward(ctx, G, z, prec):
        ctx.prec = prec
        img, saved = G._fwd(z, ctx.needs_input_grad[1], prec)
        ctx.G, ctx.saved = G, saved
        ctx.hooks = None
        if ctx.needs_input_grad[1]:              # bound to THIS forward's autograd node (stylegan2._Synthesis)
            ctx.hooks, G.bwd_hooks = G.bwd_hooks, None
        if G.debug_keep is not None and saved is not None:     # leaky-relu gates, NCHW (tests)
            G.debug_keep['gates'] = [(y > 0).permute(0, 3, 1, 2) for (_, _, y) in saved[0]]
        return img

    @staticmethod
    def backward(ctx, gimg):
        hooks, ctx.hooks = ctx.hooks, None
        return None, ctx.G._bwd(ctx.saved, gimg.contiguous(), ctx.prec, hooks), None


class Generator(nn.Module):
    def __init__(self, num_blocks=18):
        super().__init__()
        self.num_blocks = num_blocks
        self.features = nn.Sequential(*[_Block(ci, co, k, p) for (ci, co, k, p, _) in BLOCKS[:num_blocks]])
        self.output = nn.Module()
        self.output.conv = nn.Conv2d(BLOCKS[num_blocks - 1][1], 3, kernel_size=1, padding=0, bias=False)
        self.output.wscale = _WScale(3)
        for p in self.parameters():
            p.requires_grad_(False)
        self._prep = None
        self._pn_fused = {}
        self.bwd_hooks = None        # hook list for the next differentiable forward's backward (trainer.TrainStep, as stylegan2.Generator.bwd_hooks)
        self.debug_keep = None
        self.precision = 'fp32'      # arithmetic of the convs when forward() is not told otherwise (conv.PRECISION_NAMES)

    def resolve_precision(self, requested=None):
        return C.resolve(self.precision if requested is None else requested, 'proggan', 4 << ((self.num_blocks - 2) // 2))

    def _apply(self, fn, *a, **k):
        self._prep = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._prep = None
        return super().load_state_dict(*a, **k)

    def _prepare(self):
        dev = self.output.conv.weight.device
        if self._prep is not None and self._prep['dev'] == dev:
            return self._prep
        if dev.type != 'cuda':
            raise L.WgsError("ProgGAN Generator runs on the HIP kernels only: move it to the GPU (no CPU fallback)")
        P = {'dev': dev, 'layers': []}
        with torch.no_grad():
            for blk, (ci, co, k, pad, up) in zip(self.features, BLOCKS):
                wp = C.pack_weight(blk.conv.weight.float())
                wt = C.repack_w_t(wp, co, k * k, ci)
                # frozen weights: 16-bit planes (built per arithmetic mode on first use) for the patch / DMA conv kernels
                ws, wts = (C.SplitCache(wp), C.SplitCache(wt)) if (wp.numel() % 4 == 0 and ci % 16 == 0) else (None, None)      # (ci = 16: the few-channel halo kernel)
                P['layers'].append(dict(wp=wp, wt=wt, ws=ws, wts=wts, ci=ci, co=co, k=k, pad=pad, up=up,
                                        scale=float(blk.wscale.scale.item()), b=blk.wscale.b.contiguous()))
            co = self.output.conv.weight.shape[0]
            ci = self.output.conv.weight.shape[1]
            # 1x1 conv to 3 channels: output channels padded to 8 (the dgrad contraction needs Ci % 8 == 0)
            wp = torch.zeros(8, 1, ci, device=dev)
            wp[:3] = C.pack_weight(self.output.conv.weight.float())
            b = torch.zeros(8, device=dev)
            b[:3] = self.output.wscale.b
            P['out'] = dict(wp=wp, wt=C.repack_w_t(wp, 8, 1, ci), ci=ci, scale=float(self.output.wscale.scale.item()), b=b)
        self._prep, self._pn_fused = P, {}
        return P

    @staticmethod
    def _pixelnorm(x, eps=1e-8):
        y = torch.empty_like(x)
        rows, d = x.numel() // x.shape[-1], x.shape[-1]
        L.check(L.lib().wgs_pixelnorm_fwd(L.ptr(x), L.ptr(y), rows, d, L.c_float(eps), L.stream()), 'pixelnorm')
        return y

    @staticmethod
    def _pixelnorm_bwd(x, gy, eps=1e-8, act_slope=1.0, amax=None):
        """amax (a zeroed device scalar): raised to max |gx| — the magnitude bound of the fp16 operand scale of the gradient conv behind it."""
        gx = torch.empty_like(x)
        rows, d = x.numel() // x.shape[-1], x.shape[-1]
        if amax is not None:
            L.check(L.lib().wgs_pixelnorm_bwd_act_amax(L.ptr(x), L.ptr(gy), L.ptr(gx), L.ptr(amax), rows, d, L.c_float(eps), L.c_float(act_slope),
                                                       L.stream()), 'pixelnorm_bwd_amax')
        else:
            L.check(L.lib().wgs_pixelnorm_bwd_act(L.ptr(x), L.ptr(gy), L.ptr(gx), rows, d, L.c_float(eps), L.c_float(act_slope), L.stream()),
                    'pixelnorm_bwd')
        return gx

    def _fwd(self, z, save, prec):
        """z [B,512] (the wrapper reshapes to [B,512,1,1], models/gan_load.py:115-120)."""
        g = self._fwd_gen(z, save, prec, None)
        try:
            next(g)
        except StopIteration as e:
            return e.value
        raise L.WgsError("ProgGAN pass paused without a pause resolution")

    def _fwd_gen(self, z, save, prec, pause_res):
        """The pass as a Python generator (as stylegan2.Generator._synthesis_gen): yields before the first block whose output exceeds
        `pause_res` (an int or a tuple of ascending resolutions, one pause each; at least once when one is given); returns (image, saved)."""
        P = self._prepare()
        B = z.shape[0]
        x = z.contiguous().reshape(B, 1, 1, 512)
        saved = []
        pauses = [] if pause_res is None else sorted(pause_res if isinstance(pause_res, (tuple, list)) else [pause_res])
        paused = False
        for ly in P['layers']:
            H = x.shape[1] << (1 if ly['up'] else 0)
            Ho = H + 2 * ly['pad'] - ly['k'] + 1
            if pauses and Ho > pauses[0]:
                while pauses and Ho > pauses[0]:
                    pauses.pop(0)
                paused = True
                yield None
            y = torch.empty(B, Ho, Ho, ly['co'], device=z.device)
            k, pad = ly['k'], ly['pad']
            taps = [(ky - pad, kx - pad, ky * k + kx) for ky in range(k) for kx in range(k)]
            kw = dict(w_tap_stride=ly['ci'], w_row_stride=k * k * ly['ci'], ups=1 if ly['up'] else 0, alpha=ly['scale'], bias=ly['b'], act_slope=0.2,
                      gain=1.0, w_split=ly['ws'], precision=prec)
            # the 16- / 32-channel layers at 512^2 / 1024^2 in the 16-bit modes: PixelNorm of the operand inside the conv kernel (no
            # normalised tensor: 1 - 2 GB written and read back per layer at B = 32); decided once per (layer, batch, arithmetic)
            key = (id(ly), B, prec)
            if key not in self._pn_fused:
                self._pn_fused[key] = PN_FUSED and C.is_reduced(prec) and ly['ci'] in (16, 32) and C.pixelnorm_fused_ok(x, ly['wp'], y, taps, Ho, Ho, **kw)
            if self._pn_fused[key]:
                xn = None
                C.launch(x, ly['wp'], y, taps, Ho, Ho, pixelnorm_eps=1e-8, **kw)
            else:
                xn = self._pixelnorm(x)
                C.launch(xn, ly['wp'], y, taps, Ho, Ho, **kw)
            if save:
                saved.append((x, xn, y))
            x = y
        xn = self._pixelnorm(x)
        o = P['out']
        Hc = x.shape[1]
        y4 = torch.empty(B, Hc, Hc, 8, device=z.device)
        C.launch(xn, o['wp'], y4, [(0, 0, 0)], Hc, Hc, w_tap_stride=o['ci'], w_row_stride=o['ci'], alpha=o['scale'], bias=o['b'], precision=prec)
        img = y4[..., :3].permute(0, 3, 1, 2).contiguous()
        if pause_res is not None and not paused:
            yield None
        return img, ((saved, x, xn) if save else None)

    def _bwd(self, saved_all, gimg, prec, hooks=None):
        # fp16 modes: gradient operands without a magnitude bound run in split-bf16 (grad_operand=True)
        P = self._prepare()
        lib, st = L.lib(), L.stream()
        saved, x_last, xn_last = saved_all
        B = gimg.shape[0]
        dev = gimg.device
        Hc = gimg.shape[2]
        g4 = torch.zeros(B, Hc, Hc, 8, device=dev)
        g4[..., :3] = gimg.permute(0, 2, 3, 1)
        o = P['out']
        gxn = torch.empty(B, Hc, Hc, o['ci'], device=dev)
        C.launch(g4, o['wt'], gxn, [(0, 0, 0)], Hc, Hc, w_tap_stride=o['ci'] * 8, w_row_stride=8, alpha=o['scale'], precision=prec, grad_operand=True)
        # The PixelNorm input of a block is the activated output y of the block before it, so the PixelNorm backward also applies that
        # block's leaky-relu gate (y > 0 ? 1 : 0.2) while it stores: `dpre` below is d loss / d (scale * conv + b) of the block, and the
        # separate activation-backward pass over the tensor is gone.
        # fp16 modes: every PixelNorm backward also publishes max |dpre| (one atomic per wave), the magnitude bound under which the
        # gradient conv behind it rounds dpre to fp16 (power-of-two operand scale, conv_scheme.h) — without a bound those convs ran in
        # split-bf16 whatever the mode (conv._desc), i.e. a third of cfg2's generator FLOPs at three MFMAs per product
        nl = len(P['layers'])
        f16_grads = C.is_f16_operand(prec) and F16_GRADS
        amaxes = torch.zeros(nl + 1, 1, device=dev).unbind(0) if f16_grads else [None] * (nl + 1)
        dpre = self._pixelnorm_bwd(x_last, gxn, act_slope=0.2, amax=amaxes[0])
        carrier, hooks = hooks, list(hooks or ())      # [(resolution, callable)]: each called once, at the first block of <= resolution
        if carrier is not None:
            del carrier[:]
        for li, (ly, (x, xn, y)) in enumerate(zip(reversed(P['layers']), reversed(saved))):
            while hooks and max(h[0] for h in hooks) >= y.shape[1]:
                h = max(hooks, key=lambda q: q[0])
                hooks.remove(h)
                h[1]()
            k, pad = ly['k'], ly['pad']
            Hup = x.shape[1] << (1 if ly['up'] else 0)
            dup = torch.empty(B, Hup, Hup, ly['ci'], device=dev)
            taps = [(pad - ky, pad - kx, ky * k + kx) for ky in range(k) for kx in range(k)]
            C.launch(dpre, ly['wt'], dup, taps, Hup, Hup, w_tap_stride=ly['ci'] * ly['co'], w_row_stride=ly['co'], alpha=ly['scale'],
                     w_split=ly['wts'], precision=prec, grad_operand=True, **(dict(a_amax=amaxes[li], a_bound=1.0) if f16_grads else {}))
            if ly['up']:
                gxn = torch.empty_like(x)
                L.check(lib.wgs_upsample2x_bwd(L.ptr(dup), L.ptr(gxn), B, x.shape[1], x.shape[2], ly['ci'], st), 'upsample_bwd')
            else:
                gxn = dup
            # x = the previous block's activated output (gate folded in), or — first block — the latent code itself
            dpre = self._pixelnorm_bwd(x, gxn, act_slope=0.2 if li + 1 < nl else 1.0, amax=amaxes[li + 1])
        for h in sorted(hooks, key=lambda q: -q[0]):
            h[1]()
        g = dpre
        return g.reshape(B, 512)

    def forward(self, x, precision=None):
        """x: [B,512,1,1] like the reference (or [B,512]).  precision (extension): arithmetic of this call's convs."""
        return _PG.apply(self, x.reshape(x.shape[0], -1), self.resolve_precision(precision))


class ProgGANWrapper(nn.Module):
    """models/gan_load.py:109-120."""

    def __init__(self, G):
        super().__init__()
        self.G = G
        self.dim_z = 512

    @staticmethod
    def _reshape(z):
        return z.reshape(z.size()[0], z.size()[1], 1, 1)

    def forward(self, z, shift=None, precision=None):
        return self.G(self._reshape(z) if shift is None else self._reshape(z + shift), precision=precision)

    def resolve_precision(self, requested=None):
        return self.G.resolve_precision(requested)

    # -- the un-shifted pass G(z) in stages (extension; trainer.TrainStep, as StyleGAN2Wrapper) ------------------------------
    def begin(self, z, precision=None, pause_res=32):
        return L.StagedPass(self.G._fwd_gen(z.reshape(z.shape[0], -1), False, self.G.resolve_precision(precision), pause_res))

    @staticmethod
    def advance(handle):
        return handle.advance()

    @staticmethod
    def finish(handle):
        return handle.finish()


def build_proggan(pretrained_gan_weights=None, num_blocks=18):
    """models/gan_load.py:123-129 (plain state_dict file)."""
    G = Generator(num_blocks)
    if pretrained_gan_weights is not None:
        G.load_state_dict(torch.load(pretrained_gan_weights, map_location='cpu'))
    return ProgGANWrapper(G)
