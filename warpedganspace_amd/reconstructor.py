"""Reconstructor R — host-side mirror of lib/reconstructor.py:10-79 on the HIP kernels.

Same constructor (`Reconstructor(reconstructor_type, dim, channels=3)`), same `forward(x1, x2) ->
(logits [B,dim], magnitudes [B])`, same state_dict keys/shapes, so
`reconstructor.pt` / `checkpoint.pt` files are interchangeable with the reference.

ResNet branch: the reference uses torchvision's `resnet18` (un-vendored third-party dependency,
absent here); this module restates its public definition — conv7x7/2 -> BN -> ReLU -> maxpool3/2 ->
4 stages x 2 BasicBlocks (64,128,256,512) -> global avg-pool — with the first conv widened to 6 input
channels (lib/reconstructor.py:56-60).  torchvision's `fc` (512->1000) is kept as a parameter for
checkpoint compatibility but never executed: the reference runs it and discards the result
(lib/reconstructor.py:77), so neither outputs nor gradients depend on it.

Execution is an explicit kernel schedule on NHWC activations: implicit-GEMM MFMA convs (fwd / dgrad /
wgrad), fused train-mode BatchNorm(+residual)(+ReLU), max/avg pooling, small dense heads.
Arithmetic (RArith below): the reference's is fp32 everywhere, and that is what a Reconstructor runs unless told otherwise.  The
fp32-class alternative, split-bf16 x3 (3 bf16 MFMAs per product, ~2^-16 per product), can be selected separately for the forward
convs, the input-gradient convs and the weight-gradient contractions; the training step picks it when its generator runs in a
16-bit mode (TrainStep r_precision='auto').
"""
from collections import namedtuple

import torch
from torch import nn

from . import _lib as L
from . import conv as C

BN_EPS, BN_MOM = 1e-5, 0.1


class RArith(namedtuple('RArith', ['forward', 'dgrad', 'wgrad'])):
    """Arithmetic of the Reconstructor's convs: 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = split-bf16 x3 (fp32-class),
    5 = fp32 with the 3x3 stride-1 convs in Winograd form (conv.FP32W; forward / dgrad only).
      forward  the forward convs.  ReLU gates and train-mode BatchNorm statistics are fixed by the forward; on identical inputs the
               extra gate flips of split-bf16 move single parameter-gradient entries by ~2e-2 of the tensor maximum (< 1e-3 exact,
               tests/test_reconstructor_gpu.py) - but inside a step whose generator runs in a 16-bit mode R never sees identical
               inputs: its images differ from the fp32 ones by ~5e-4, which flips far more gates than a 1e-5 perturbation of R's
               own convs (and the reference's own convs run in TF32, 2^-11, on the GPUs it targets);
      dgrad    the BasicBlocks' and conv1's input-gradient convs (linear in dy for fixed gates / statistics: a smooth ~1e-5);
      wgrad    the weight-gradient contractions where the 16-bit kernels beat the exact one (>= 128 channels on both sides, and
               the stride-1 3x3 convs at 64 channels through the kernel-row form, conv_wgrad16.hip); the others stay exact."""


R_EXACT = RArith(0, 0, 0)            # the reference's arithmetic
R_FP32_CLASS = RArith(1, 1, 1)
R_FP32_WINO = RArith(5, 5, 0)        # fp32; the 3x3 stride-1 forward / input-gradient convs in Winograd F(2x2,3x3) form (conv.FP32W)


def r_arith(r_precision='auto', generator_code=None):
    """RArith for a requested mode: 'fp32' exact everywhere; 'bf16x3' fp32-class everywhere; 'auto' follows the generator the
    step runs (exact when it is exact fp32 or unknown, fp32-class when it runs in any 16-bit mode); an RArith passes through."""
    if isinstance(r_precision, RArith):
        return r_precision
    key = str(r_precision).lower()
    if key in ('fp32', '0'):
        return R_EXACT
    if key in ('fp32w', '5'):
        return R_FP32_WINO
    if key in ('bf16x3', '1'):
        return R_FP32_CLASS
    if key == 'auto':
        if generator_code == 5:
            return R_FP32_WINO
        return R_FP32_CLASS if (generator_code is not None and generator_code >= 1) else R_EXACT
    raise L.WgsError("unknown reconstructor precision %r (fp32, fp32w, bf16x3, auto)" % (r_precision,))


# The ResNet stem in space-to-depth form (Reconstructor._forward_impl): taps (dy, dx, weight index r*4 + s) of the 4 x 4 block window
BN_FUSED_APPLY = True         # round 5: ... and the finalise / collapse launches folded into the apply kernels' prologues (_BNScratch)
BN_EPILOGUE_STATS = True      # round 5: BatchNorm statistics from the producing conv's epilogue (wgs_conv_desc.col_stats), see _forward_impl
STEM_S2D = True
STEM_WGRAD_S2D = True     # ... and its weight gradient in the same form (64 x 16 x 32, gathered back to 64 x 49 x 2c)
_S2D_TAPS = [(r - 2, s_ - 2, r * 4 + s_) for r in range(4) for s_ in range(4)]


def _conv(ci, co, k, stride, pad):
    m = nn.Conv2d(ci, co, k, stride=stride, padding=pad, bias=False)
    nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
    return m


class _BasicBlock(nn.Module):
    """Parameter container with torchvision BasicBlock's names (conv1, bn1, conv2, bn2, downsample.{0,1})."""

    def __init__(self, ci, co, stride):
        super().__init__()
        self.conv1 = _conv(ci, co, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(co)
        self.conv2 = _conv(co, co, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(co)
        self.stride = stride
        if stride != 1 or ci != co:
            self.downsample = nn.Sequential(_conv(ci, co, 1, stride, 0), nn.BatchNorm2d(co))
        else:
            self.downsample = None


class _ResNet18(nn.Module):
    """Parameter container with torchvision resnet18's names."""

    def __init__(self, in_channels=6):
        super().__init__()
        self.conv1 = _conv(in_channels, 64, 7, 2, 3)
        self.bn1 = nn.BatchNorm2d(64)
        chans = [64, 128, 256, 512]
        ci = 64
        for i, co in enumerate(chans):
            stride = 1 if i == 0 else 2
            setattr(self, 'layer%d' % (i + 1), nn.Sequential(_BasicBlock(ci, co, stride), _BasicBlock(co, co, 1)))
            ci = co
        self.fc = nn.Linear(512, 1000)   # present in the reference's state_dict; never executed (see module doc)

    def blocks(self):
        return [b for i in range(1, 5) for b in getattr(self, 'layer%d' % i)]


def _packed(conv):
    """[Co, T, Ci] view of a conv weight whose memory is channels_last (re-laid out once if it is not)."""
    w = conv.weight
    Co, Ci, kh, kw = w.shape
    if not w.permute(0, 2, 3, 1).is_contiguous():
        w.data = w.data.contiguous(memory_format=torch.channels_last)
        if not w.permute(0, 2, 3, 1).is_contiguous():   # 1x1 / degenerate strides: force an explicit copy
            w.data = w.data.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return w.detach().permute(0, 2, 3, 1).reshape(Co, kh * kw, Ci)


def _grad_like(conv, dw_packed):
    """Logical [Co,Ci,kh,kw] (channels_last strided) view of a packed [Co,T,Ci] gradient."""
    Co, Ci, kh, kw = conv.weight.shape
    return dw_packed.view(Co, kh, kw, Ci).permute(0, 3, 1, 2)


class _BNScratch:
    """The PAIR of fp64 scratch buffers of the fused BatchNorm launches (wgs_bn_fwd_fused / wgs_bn_bwd_fused): a producer (a conv's
    epilogue, the backward's reduction) accumulates into cur(), the fused apply launch behind it reads that buffer and leaves the OTHER
    one zero, and the roles swap.  Both zero at the start; the index persists across steps (the buffer a step leaves dirty is the one its
    successor's first apply launch zeroes)."""

    def __init__(self, dev, C=512):
        self.buf = [torch.zeros(64 * C, dtype=torch.float64, device=dev) for _ in range(2)]
        self.k = 0

    def cur(self):
        return self.buf[self.k]

    def consume(self):
        """(buffer holding the producer's sums, buffer to leave zero); swaps the roles."""
        a, b = self.buf[self.k], self.buf[self.k ^ 1]
        self.k ^= 1
        return a, b

    def reset(self):
        for b in self.buf:
            b.zero_()
        self.k = 0


class _BN:
    """One fused BatchNorm launch pair + what its backward needs."""

    @staticmethod
    def fwd(bn, x, ws, residual=None, relu=True, train=True, sums_ready=False):
        """sums_ready: the conv that produced x accumulated its column sums into ws from its epilogue (col_stats=ws): no statistics launch."""
        N, Cn = x.numel() // x.shape[-1], x.shape[-1]
        y = torch.empty_like(x)
        mean = torch.empty(Cn, device=x.device)
        invstd = torch.empty(Cn, device=x.device)
        if sums_ready and train:
            if isinstance(ws, _BNScratch):      # one launch: statistics finished in the apply kernel's prologue, scratch pair
                a, b = ws.consume()
                L.check(L.lib().wgs_bn_fwd_fused(L.ptr(x), L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(residual), L.ptr(y), L.ptr(mean),
                                                 L.ptr(invstd), L.ptr(bn.running_mean), L.ptr(bn.running_var),
                                                 L.ptr(bn.num_batches_tracked, torch.int64), L.rawptr(a), L.rawptr(b), L.c_int64(N), Cn,
                                                 L.c_float(bn.eps), L.c_float(bn.momentum if bn.momentum is not None else 0.1),
                                                 int(relu), L.stream()), 'wgs_bn_fwd_fused')
                return y, (mean, invstd)
            L.check(L.lib().wgs_bn_fwd_sums(L.ptr(x), L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(residual), L.ptr(y), L.ptr(mean),
                                            L.ptr(invstd), L.ptr(bn.running_mean), L.ptr(bn.running_var),
                                            L.ptr(bn.num_batches_tracked, torch.int64), L.rawptr(ws), L.c_int64(N), Cn,
                                            L.c_float(bn.eps), L.c_float(bn.momentum if bn.momentum is not None else 0.1),
                                            int(relu), L.stream()), 'wgs_bn_fwd_sums')
            return y, (mean, invstd)
        if isinstance(ws, _BNScratch):
            ws = ws.cur()                       # (no sums from an epilogue: the three-launch form on the buffer that is zero, left zero)
        L.check(L.lib().wgs_bn_fwd(L.ptr(x), L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(residual), L.ptr(y), L.ptr(mean),
                                   L.ptr(invstd), L.ptr(bn.running_mean), L.ptr(bn.running_var),
                                   L.ptr(bn.num_batches_tracked, torch.int64), L.rawptr(ws), L.c_int64(N), Cn,
                                   L.c_float(bn.eps), L.c_float(bn.momentum if bn.momentum is not None else 0.1),
                                   int(relu), int(train), L.stream()), 'wgs_bn_fwd')
        for _ in range(_BN.debug_extra_launches):      # development (tools/ab_tail.py): what does a launch in R's chain cost by itself?
            L.check(L.lib().wgs_split_bf16(L.ptr(mean[:4]), L.ptr(_BN._dummy(x.device), torch.int16), L.ptr(_BN._dummy(x.device)[4:], torch.int16),
                                           L.c_int64(4), L.stream()), 'dummy')
        return y, (mean, invstd)

    debug_extra_launches = 0
    _dummy_buf = {}

    @staticmethod
    def _dummy(dev):
        if dev not in _BN._dummy_buf:
            _BN._dummy_buf[dev] = torch.zeros(8, dtype=torch.int16, device=dev)
        return _BN._dummy_buf[dev]

    @staticmethod
    def bwd(bn, x, stats, dyA, dyB, out, ws, want_res=False, train=True, gbuf=None):
        N, Cn = x.numel() // x.shape[-1], x.shape[-1]
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if want_res else None
        dg = gbuf[id(bn.weight)] if gbuf is not None else torch.empty(Cn, device=x.device)
        db = gbuf[id(bn.bias)] if gbuf is not None else torch.empty(Cn, device=x.device)
        if isinstance(ws, _BNScratch):
            if train:                           # two launches instead of three: reduction, then the apply kernel finishes the sums itself
                a, b = ws.consume()
                L.check(L.lib().wgs_bn_bwd_fused(L.ptr(x), L.ptr(dyA), L.ptr(dyB), L.ptr(out), L.ptr(stats[0]), L.ptr(stats[1]),
                                                 L.ptr(bn.weight), L.ptr(dx), L.ptr(dres), L.ptr(dg), L.ptr(db), L.rawptr(a), L.rawptr(b),
                                                 L.c_int64(N), Cn, L.stream()), 'wgs_bn_bwd_fused')
                return dx, dres, dg, db
            ws = ws.cur()                       # (eval-mode backward: the three-launch form on the buffer that is zero)
        L.check(L.lib().wgs_bn_bwd(L.ptr(x), L.ptr(dyA), L.ptr(dyB), L.ptr(out), L.ptr(stats[0]), L.ptr(stats[1]),
                                   L.ptr(bn.weight), L.ptr(dx), L.ptr(dres), L.ptr(dg), L.ptr(db), L.rawptr(ws),
                                   L.c_int64(N), Cn, int(train), L.stream()), 'wgs_bn_bwd')
        return dx, dres, dg, db


class StepWeights:
    """Per-step derived forms of the Reconstructor's trained conv weights, owned by a training engine (trainer.TrainStep): the [T, Ci, Co]
    copies its input-gradient convs contract with and, in 'fp32w', the Winograd U operands of its forward and input-gradient convs.
    refresh() re-derives all of them from the parameters' current values on the current stream — the engine calls it once per step, after
    the previous Adam update, on its side stream — so that none of these ~60 small launches sits in the critical chain of R's forward /
    backward.  Pass the object to _forward_impl / _backward_impl (sw=) ONLY in steps that refreshed it."""

    def __init__(self, R):
        self.R, self.caches, self.wt = R, {}, {}

    def cache(self, conv, kind):
        return self.caches.setdefault((id(conv), kind), C.StepWinoCache())

    def refresh(self):
        self.wt = self.R.prepare_dgrad_weights() or {}
        for c in self.caches.values():
            c.refresh()


class _RFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R, x1, x2, *params):
        logits, mag, saved = R._forward_impl(x1, x2, save=any(ctx.needs_input_grad))
        ctx.R, ctx.saved = R, saved
        ctx.need_x = (ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return logits, mag

    @staticmethod
    def backward(ctx, dlogits, dmag):
        R = ctx.R
        grads, d1, d2 = R._backward_impl(ctx.saved, dlogits.contiguous(), dmag.contiguous(), ctx.need_x)
        plist = [grads.get(id(p)) if p.requires_grad else None for p in R._param_list()]
        return (None, d1, d2) + tuple(plist)


class Reconstructor(nn.Module):
    def __init__(self, reconstructor_type, dim, channels=3):
        super().__init__()
        self.reconstructor_type = reconstructor_type
        self.dim = dim
        self.channels = channels
        if reconstructor_type == 'ResNet':
            self.features_extractor = _ResNet18(in_channels=2 * channels)
            self.path_indices = nn.Linear(512, dim)          # lib/reconstructor.py:66
            self.shift_magnitudes = nn.Linear(512, 1)        # :69
        elif reconstructor_type == 'LeNet':
            from .lenet import build_lenet
            build_lenet(self)
        else:
            raise ValueError("reconstructor_type must be 'ResNet' or 'LeNet', got %r" % (reconstructor_type,))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        self.arith = R_EXACT         # arithmetic of forward() / _forward_impl() calls that do not pass one

    def _param_list(self):
        return list(self.parameters())

    # -- persistent per-device scratch (allocated once, not per step) -----------------------------------------------
    def _scratch(self, name, shape, dtype, dev, zero=False):
        store = self.__dict__.setdefault('_scratch_bufs', {})
        key = (name, tuple(shape), dtype, str(dev))
        if key not in store:
            store[key] = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=dev)
        return store[key]

    def _conv1_padded(self, c, Cp, dev):
        """conv1's packed weight [64, 49, 2c] widened to Cp input channels (zero padding written once; the live 2c channels are
        refreshed from the parameter on every call — one small strided copy instead of a zero-fill + copy)."""
        w1p = self._scratch('w1p', (64, 49, Cp), torch.float32, dev, zero=True)
        w1p[:, :, :2 * c].copy_(_packed(self.features_extractor.conv1))
        return w1p

    # -- transposed weights for the input-gradient convs, ahead of time -----------------------------------------------
    def prepare_dgrad_weights(self):
        """[T, Ci, Co] copies of every block conv's weights (what conv2d_dgrad contracts with) into persistent scratch, enqueued on the
        current stream: {id(conv): tensor} for _backward_impl(wt=...).  The weights of a training step are fixed from the previous
        Adam update on, so TrainStep issues these 20 small launches at the start of the step on its side stream instead of one by
        one inside R's backward, where each sits in the critical path's chain of dependent launches."""
        if self.reconstructor_type != 'ResNet':
            return None
        out = {}
        for blk in self.features_extractor.blocks():
            for conv in (blk.conv1, blk.conv2) + ((blk.downsample[0],) if blk.downsample is not None else ()):
                w = _packed(conv)
                Co, T, Ci = w.shape
                out[id(conv)] = C.repack_w_t(w, Co, T, Ci, out=self._scratch('wt%d' % id(conv), (T, Ci, Co), torch.float32, w.device))
        return out

    # -- reference signature ---------------------------------------------------------------------------
    def forward(self, x1, x2):
        if not x1.is_cuda:
            raise L.WgsError("Reconstructor runs on the HIP kernels only: inputs must be GPU tensors (no CPU fallback)")
        return _RFunction.apply(self, x1, x2, *self._param_list())

    # -- explicit schedule -------------------------------------------------------------------------------
    def _forward_impl(self, x1, x2, save=True, arith=None, sw=None):
        """The forward schedule.  An exception anywhere in it (out of memory, a WgsError, KeyboardInterrupt) may leave a BatchNorm scratch buffer
        holding a conv's sums with no apply launch behind it: the scratch pairs are cleared before it propagates, so that a later step does
        not silently compute its statistics on top of them."""
        try:
            return self._forward_body(x1, x2, save, arith, sw)
        except BaseException:
            self._reset_bn_scratch()
            raise

    def _reset_bn_scratch(self):
        for pair in self.__dict__.get('_bn_pairs', {}).values():
            try:
                pair.reset()
            except Exception:  # noqa: BLE001  (a device that is gone: nothing left to protect)
                pass

    def _forward_body(self, x1, x2, save=True, arith=None, sw=None):
        if self.reconstructor_type == 'LeNet':
            from . import lenet
            return lenet.forward_impl(self, x1, x2, save)
        fe = self.features_extractor
        train = self.training
        lib, st = L.lib(), L.stream()
        dev = x1.device
        x1, x2 = x1.contiguous(), x2.contiguous()
        B, c, H, W = x1.shape
        ws = self._scratch('bn_ws', (64 * 512,), torch.float64, dev, zero=True)      # WGS_BN_WS_DOUBLES(512): zero on entry, left zero
        Cp = 8
        arith = arith or self.arith
        fp = arith.forward
        # train-mode BatchNorm statistics out of the producing conv's epilogue (wgs_conv_desc.col_stats) wherever the conv runs a tiled
        # kernel with the shared epilogue or the Winograd kernel: every conv of the direct fp32, 'fp32w' and split-bf16 arithmetics.  One scratch: a conv's sums are consumed (and the scratch left zero) by the BatchNorm right behind it.
        est = BN_EPILOGUE_STATS and train and fp in (0, 1, C.FP32W)
        if est and BN_FUSED_APPLY:
            # ... and finished in the apply kernel's prologue over a scratch PAIR (_BNScratch): conv -> apply, one BatchNorm launch instead of
            # three; the backward's three launches become two the same way
            # one pair per (device, stream): two engines / streams driving the same Reconstructor do not share accumulators
            pairs = self.__dict__.setdefault('_bn_pairs', {})
            key = (dev.index, L.stream_id(dev.index))
            pair = pairs.get(key)
            if pair is None:
                pair = pairs[key] = _BNScratch(dev)
            ws = pair
        cs = (lambda: dict(col_stats=ws.cur() if isinstance(ws, _BNScratch) else ws)) if est else (lambda: {})
        # The stem (7 x 7, stride 2, 2c = 6 input channels): SPACE-TO-DEPTH.  The image pair is packed as
        # [B, H/2, W/2, 32] (2 x 2 pixel block x 8 channels) and the 7 x 7 / 2 conv becomes a 4 x 4-window stride-1 conv 32 -> 64 channels
        # over it (zeros where a tap falls outside the 7 x 7: 49 * 6 of 16 * 32 weights live) — a shape the few-channel halo kernel
        # (conv_halo16.hip) runs HBM-bound in the 16-bit modes, instead of a GEMM whose rows gather 49 taps of 6 channels.  Same products,
        # same roundings.  In exact fp32 the two launches go through the fp32 template (conv_igemm_f32.hip): the forward costs what the
        # gather form did, the image gradient — 64 -> 8 channels in four stride phases of 128 x 32 tiles before — a third (0.95 -> 0.34 ms).
        s2d = STEM_S2D and H % 16 == 0 and W % 64 == 0 and 2 * c <= 8
        if s2d:
            x = torch.empty(B, H // 2, W // 2, 32, device=dev)
            L.check(lib.wgs_pack_pair_s2d(L.ptr(x1), L.ptr(x2), L.ptr(x), B, c, H, W, st), 'pack_pair_s2d')
            w1s = self._scratch('w1s', (64, 16, 32), torch.float32, dev)
            L.check(lib.wgs_stem_weight_s2d(L.rawptr(_packed(fe.conv1)), L.ptr(w1s), 64, 2 * c, 0, st), 'stem_weight_s2d')
            c1 = torch.empty(B, H // 2, W // 2, 64, device=dev)
            C.launch(x, w1s, c1, _S2D_TAPS, H // 2, W // 2, w_tap_stride=32, w_row_stride=16 * 32, precision=fp, **cs())
        else:
            x = torch.empty(B, H, W, Cp, device=dev)
            L.check(lib.wgs_pack_pair_nhwc(L.ptr(x1), L.ptr(x2), L.ptr(x), B, c, H * W, Cp, st), 'pack_pair')
            # stem: conv1 weights padded from 2c to Cp input channels
            w1p = self._conv1_padded(c, Cp, dev)
            c1 = C.conv2d(x, w1p, 7, stride=2, pad=3, precision=fp)
        a1, st1 = _BN.fwd(fe.bn1, c1, ws, relu=True, train=train, sums_ready=est and s2d)
        Hp = (a1.shape[1] + 2 - 3) // 2 + 1
        p1 = torch.empty(B, Hp, Hp, 64, device=dev)
        idx = torch.empty(B, Hp, Hp, 64, dtype=torch.uint8, device=dev)
        L.check(lib.wgs_maxpool_fwd(L.ptr(a1), L.ptr(p1), L.rawptr(idx), B, a1.shape[1], a1.shape[2], 64, 3, 2, 1, st),
                'maxpool')
        saved_blocks = []
        h = p1
        wc = (lambda conv: sw.cache(conv, 'f')) if sw is not None else (lambda conv: None)       # Winograd operands refreshed by the engine (StepWeights)
        for blk in fe.blocks():
            xin = h
            ca = C.conv2d(xin, _packed(blk.conv1), 3, stride=blk.stride, pad=1, precision=fp, w_split=wc(blk.conv1), **cs())
            aa, sa = _BN.fwd(blk.bn1, ca, ws, relu=True, train=train, sums_ready=est)
            if blk.downsample is not None:       # (before conv2: one scratch, consumed by the BatchNorm right behind each conv)
                cd = C.conv2d(xin, _packed(blk.downsample[0]), 1, stride=blk.stride, pad=0, precision=fp, **cs())
                ad, sd = _BN.fwd(blk.downsample[1], cd, ws, relu=False, train=train, sums_ready=est)
                ident = ad
            else:
                cd, sd, ident = None, None, xin
            cb = C.conv2d(aa, _packed(blk.conv2), 3, stride=1, pad=1, precision=fp, w_split=wc(blk.conv2), **cs())
            o, sb = _BN.fwd(blk.bn2, cb, ws, residual=ident, relu=True, train=train, sums_ready=est)
            saved_blocks.append((xin, ca, aa, sa, cb, sb, cd, sd, o))
            h = o
        P = h.shape[1] * h.shape[2]
        feat = torch.empty(B, 512, device=dev)
        L.check(lib.wgs_avgpool_fwd(L.ptr(h), L.ptr(feat), B, P, 512, st), 'avgpool')
        K = self.dim
        logits = torch.empty(B, K, device=dev)
        mag = torch.empty(B, 1, device=dev)
        for lin, out, n in ((self.path_indices, logits, K), (self.shift_magnitudes, mag, 1)):
            L.check(lib.wgs_linear_fwd(L.ptr(feat), L.ptr(lin.weight), L.ptr(lin.bias), L.ptr(out), B, n, 512, 512, n,
                                       L.c_float(1.0), L.c_float(1.0), 0, 0, L.c_float(0.0), L.c_float(1.0), st), 'head')
        saved = dict(x=x, c1=c1, a1=a1, st1=st1, idx=idx, p1shape=p1.shape, blocks=saved_blocks, feat=feat, hshape=h.shape,
                     B=B, c=c, H=H, W=W, Cp=Cp, train=train, ws=ws, arith=arith, s2d=s2d) if save else None
        return logits, mag.reshape(B) if B > 1 else mag.squeeze(), saved

    def _backward_impl(self, S, dlogits, dmag, need_x=(False, True), gbuf=None, deferred=None, wt=None, sw=None):
        try:
            return self._backward_body(S, dlogits, dmag, need_x, gbuf, deferred, wt, sw)
        except BaseException:
            self._reset_bn_scratch()          # (see _forward_impl)
            raise

    def _backward_body(self, S, dlogits, dmag, need_x=(False, True), gbuf=None, deferred=None, wt=None, sw=None):
        """Returns ({id(param): grad}, d_x1 or None, d_x2 or None).  With `gbuf` ({id(param): zero-initialised
        buffer in the parameter's MEMORY layout, conv weights packed [Co,T,Ci]}) gradients are written /
        accumulated straight into those buffers (the trainer's flat gradient bucket)."""
        if self.reconstructor_type == 'LeNet':
            from . import lenet
            return lenet.backward_impl(self, S, dlogits, dmag, need_x, gbuf)
        fe = self.features_extractor
        lib, st = L.lib(), L.stream()
        B, train, ws = S['B'], S['train'], S['ws']
        arith = S['arith']           # the backward runs in the arithmetic the forward was told
        dev = dlogits.device
        K = self.dim
        grads = {}

        # Weight gradients are not on the path to the image gradient.  With `deferred` (a list) they are queued as closures
        # and the caller runs them where it likes (TrainStep: on a side stream, next to the generator's backward).
        def wgrad(x, dy, dw, k, stride, pad):
            # split-bf16 where it beats the exact kernel: >= 128 channels on both sides, and the stride-1 3x3 convs at 64
            # channels through the kernel-row form (conv_wgrad16.hip: 88 vs 65 TFLOP/s)
            prec = arith.wgrad if (min(x.shape[-1], dy.shape[-1]) >= 128 or (k == 3 and stride == 1 and x.shape[-1] % 64 == 0)) else 0
            if deferred is None:
                C.conv2d_wgrad(x, dy, dw, k, stride=stride, pad=pad, precision=prec)
            else:
                deferred.append((x, dy, lambda: C.conv2d_wgrad(x, dy, dw, k, stride=stride, pad=pad, precision=prec)))

        if sw is not None and wt is None:
            wt = sw.wt

        def w_t(conv, w, Co, T, Ci):          # the conv's weights as [T, Ci, Co]: prepared ahead (prepare_dgrad_weights) or made here
            return wt[id(conv)] if (wt is not None and id(conv) in wt) else C.repack_w_t(w, Co, T, Ci)

        def wcd(conv):                        # Winograd operand cache of the conv's input-gradient launch: only beside a prepared [T, Ci, Co] copy
            return sw.cache(conv, 'd') if (sw is not None and wt is not None and id(conv) in wt) else None

        feat = S['feat']
        dmag = dmag.reshape(B, 1)
        dfeat = torch.empty(B, 512, device=dev)
        for i, (lin, g, n) in enumerate(((self.path_indices, dlogits, K), (self.shift_magnitudes, dmag, 1))):
            L.check(lib.wgs_linear_dgrad(L.ptr(g), L.ptr(lin.weight), None, L.ptr(dfeat), B, n, 512, n, 512, L.c_float(1.0),
                                         L.c_float(1.0), L.c_float(1.0), i, st), 'head_dgrad')
            if gbuf is not None:
                dw, db = gbuf[id(lin.weight)], gbuf[id(lin.bias)]
            else:
                dw, db = torch.empty_like(lin.weight), torch.empty_like(lin.bias)
            L.check(lib.wgs_linear_wgrad(L.ptr(g), L.ptr(feat), L.ptr(dw), L.ptr(db), B, n, 512, st), 'head_wgrad')
            grads[id(lin.weight)], grads[id(lin.bias)] = dw, db
        hs = S['hshape']
        dh = torch.empty(hs, device=dev)
        L.check(lib.wgs_avgpool_bwd(L.ptr(dfeat), L.ptr(dh), B, hs[1] * hs[2], 512, st), 'avgpool_bwd')
        dyA, dyB = dh, None
        blocks = fe.blocks()
        for blk, (xin, ca, aa, sa, cb, sb, cd, sd, o) in zip(reversed(blocks), reversed(S['blocks'])):
            # o = relu(bn2(cb) + identity)
            dcb, dres, dg, db_ = _BN.bwd(blk.bn2, cb, sb, dyA, dyB, o, ws, want_res=True, train=train, gbuf=gbuf)
            grads[id(blk.bn2.weight)], grads[id(blk.bn2.bias)] = dg, db_
            w2 = _packed(blk.conv2)
            Co, T, Ci = w2.shape
            dw2 = gbuf[id(blk.conv2.weight)] if gbuf is not None else torch.zeros_like(w2)
            wgrad(aa, dcb, dw2, 3, 1, 1)
            grads[id(blk.conv2.weight)] = _grad_like(blk.conv2, dw2)
            daa = C.conv2d_dgrad(dcb, w_t(blk.conv2, w2, Co, T, Ci), aa.shape[1:3], 3, stride=1, pad=1, precision=arith.dgrad, w_split=wcd(blk.conv2))
            dca, _, dg, db_ = _BN.bwd(blk.bn1, ca, sa, daa, None, aa, ws, train=train, gbuf=gbuf)
            grads[id(blk.bn1.weight)], grads[id(blk.bn1.bias)] = dg, db_
            w1 = _packed(blk.conv1)
            Co, T, Ci = w1.shape
            dw1 = gbuf[id(blk.conv1.weight)] if gbuf is not None else torch.zeros_like(w1)
            wgrad(xin, dca, dw1, 3, blk.stride, 1)
            grads[id(blk.conv1.weight)] = _grad_like(blk.conv1, dw1)
            dmain = C.conv2d_dgrad(dca, w_t(blk.conv1, w1, Co, T, Ci), xin.shape[1:3], 3, stride=blk.stride, pad=1, precision=arith.dgrad, w_split=wcd(blk.conv1))
            if blk.downsample is not None:
                dcd, _, dg, db_ = _BN.bwd(blk.downsample[1], cd, sd, dres, None, None, ws, train=train, gbuf=gbuf)
                grads[id(blk.downsample[1].weight)], grads[id(blk.downsample[1].bias)] = dg, db_
                wd = _packed(blk.downsample[0])
                Co, T, Ci = wd.shape
                dwd = gbuf[id(blk.downsample[0].weight)] if gbuf is not None else torch.zeros_like(wd)
                wgrad(xin, dcd, dwd, 1, blk.stride, 0)
                grads[id(blk.downsample[0].weight)] = _grad_like(blk.downsample[0], dwd)
                dside = C.conv2d_dgrad(dcd, w_t(blk.downsample[0], wd, Co, T, Ci), xin.shape[1:3], 1, stride=blk.stride, pad=0, precision=arith.dgrad)
            else:
                dside = dres
            dyA, dyB = dmain, dside
        # stem
        dp1 = dyA + dyB
        a1 = S['a1']
        da1 = torch.empty_like(a1)
        L.check(lib.wgs_maxpool_bwd(L.ptr(dp1), L.rawptr(S['idx']), L.ptr(da1), B, a1.shape[1], a1.shape[2], 64, 3, 2, 1, st),
                'maxpool_bwd')
        dc1, _, dg, db_ = _BN.bwd(fe.bn1, S['c1'], S['st1'], da1, None, a1, ws, train=train, gbuf=gbuf)
        grads[id(fe.bn1.weight)], grads[id(fe.bn1.bias)] = dg, db_
        c, Cp = S['c'], S['Cp']
        # the stem's weight gradient is not on the path to the image gradient either (0.45 ms in front of it at 256^2 inputs): with
        # `deferred` it joins the other weight gradients; the caller's flat bucket (gbuf) receives it when that closure runs
        x_stem, s2d_stem = S['x'], S['s2d']
        s2d_form = s2d_stem and STEM_WGRAD_S2D and 2 * c <= 8
        # s2d input: the gradient is taken in the s2d form too — a 4 x 4-window stride-1 conv over 32 channels, 512 (tap, channel)
        # columns of which 49 * 2c are live: 30 % more products than the 7 x 7 form, but a pixel's 32 channels are one 128-byte line
        # instead of 49 scattered 32-byte pieces (the 7 x 7 form was bound by its gather, 54-61 TFLOP/s in fp32 AND in split-bf16)
        dw1p = torch.zeros(64, 16, 32, device=dev) if s2d_form else torch.zeros(64, 49, Cp, device=dev)

        def stem_wgrad():
            if s2d_form:
                C.conv2d_wgrad(x_stem, dc1, dw1p, 4, stride=1, pad=2, precision=arith.wgrad)
                dst = gbuf[id(fe.conv1.weight)] if gbuf is not None else torch.empty(64, 49, 2 * c, device=dev)
                L.check(L.lib().wgs_stem_weight_s2d(L.ptr(dw1p), L.rawptr(dst), 64, 2 * c, 1, L.stream()), 'stem_weight_s2d(back)')
                res = dst
            else:
                # (reads the s2d input in place; split-bf16 where the arithmetic allows it: 392 flattened (tap, channel) columns)
                C.conv2d_wgrad(x_stem, dc1, dw1p, 7, stride=2, pad=3, x_s2d=s2d_stem, precision=arith.wgrad)
                if gbuf is not None:
                    gbuf[id(fe.conv1.weight)].copy_(dw1p[:, :, :2 * c])
                res = dw1p[:, :, :2 * c]
            if dw1p.is_cuda:
                dw1p.record_stream(torch.cuda.current_stream(dw1p.device))     # (allocated on the caller's stream, used on the deferred one)
            return res
        if deferred is None or gbuf is None:
            grads[id(fe.conv1.weight)] = _grad_like(fe.conv1, stem_wgrad().contiguous())
        else:
            deferred.append((x_stem, dc1, stem_wgrad))
            grads[id(fe.conv1.weight)] = _grad_like(fe.conv1, gbuf[id(fe.conv1.weight)].view(64, 49, 2 * c))
        d1 = d2 = None
        if (need_x[0] or need_x[1]) and S['s2d']:
            # image gradient in the space-to-depth form: 64 -> 32 channels over the transposed 4 x 4 window, then depth-to-space
            w1s = self._scratch('w1s', (64, 16, 32), torch.float32, dev)      # this step's forward weights (Adam runs after the backward)
            w1st = C.repack_w_t(w1s, 64, 16, 32, out=self._scratch('w1st', (16, 32, 64), torch.float32, dev))
            dxs = torch.empty(B, S['H'] // 2, S['W'] // 2, 32, device=dev)
            C.launch(dc1, w1st, dxs, [(-dy, -dx, t) for dy, dx, t in _S2D_TAPS], S['H'] // 2, S['W'] // 2, w_tap_stride=32 * 64, w_row_stride=64,
                     precision=arith.dgrad, grad_operand=True)
            d1 = torch.empty(B, c, S['H'], S['W'], device=dev) if need_x[0] else None
            d2 = torch.empty(B, c, S['H'], S['W'], device=dev) if need_x[1] else None
            L.check(lib.wgs_unpack_pair_s2d_grad(L.ptr(dxs), L.ptr(d1), L.ptr(d2), B, c, S['H'], S['W'], st), 'unpack_pair_s2d')
        elif need_x[0] or need_x[1]:
            w1p = self._conv1_padded(c, Cp, dev)        # same weights as in the forward of this step (Adam runs after the backward)
            w1t = C.repack_w_t(w1p, 64, 49, Cp, out=self._scratch('w1t', (49, Cp, 64), torch.float32, dev))
            dx = C.conv2d_dgrad(dc1, w1t, (S["H"], S["W"]), 7, stride=2, pad=3, precision=arith.dgrad)
            d1 = torch.empty(B, c, S['H'], S['W'], device=dev) if need_x[0] else None
            d2 = torch.empty(B, c, S['H'], S['W'], device=dev) if need_x[1] else None
            L.check(lib.wgs_unpack_pair_grad(L.ptr(dx), L.ptr(d1), L.ptr(d2), B, c, S['H'] * S['W'], Cp, st), 'unpack_pair')
        return grads, d1, d2
