"""Host-side launch helpers for the implicit-GEMM convolution kernels (wgs_conv_igemm /
wgs_conv_wgrad / wgs_repack_w_t).  All activations are NHWC tensors of shape [B, H, W, C];
weights are "packed" as [Cout, T, Cin] (T = kh*kw taps; this is the memory of a PyTorch
[Cout, Cin, kh, kw] tensor in channels_last format) or, for dgrad contractions, [T, Cin, Cout].
"""
import ctypes

import torch

from . import _lib as L


class ConvDesc(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('w', ctypes.c_void_p), ('y', ctypes.c_void_p),
                ('a_scale', ctypes.c_void_p), ('col_scale', ctypes.c_void_p), ('bias', ctypes.c_void_p),
                ('noise', ctypes.c_void_p), ('noise_w', ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ('B', 'Hi', 'Wi', 'Ci', 'Hg', 'Wg', 'isy', 'isx', 'Ho', 'Wo', 'Co',
                                              'osy', 'osx', 'oy0', 'ox0', 'ntaps', 'a_ld', 'col_ld', 'ups', 'add_ups',
                                              'act', 'precision')] + \
               [('alpha', ctypes.c_float), ('addend', ctypes.c_void_p),
                ('w_tap_stride', ctypes.c_int64), ('w_row_stride', ctypes.c_int64),
                ('act_slope', ctypes.c_float), ('gain', ctypes.c_float),
                ('dy', ctypes.c_int8 * 64), ('dx', ctypes.c_int8 * 64), ('wt', ctypes.c_int16 * 64),
                ('w_hi', ctypes.c_void_p), ('w_lo', ctypes.c_void_p),
                ('ws', ctypes.c_void_p), ('ws_bytes', ctypes.c_int64),
                ('a_amax', ctypes.c_void_p), ('a_bound', ctypes.c_float), ('a_amax2', ctypes.c_void_p), ('y_amax', ctypes.c_void_p),
                ('x_f16', ctypes.c_void_p),
                ('rgb_out', ctypes.c_void_p), ('rgb_s', ctypes.c_void_p), ('rgb_w', ctypes.c_void_p), ('rgb_scale', ctypes.c_float), ('rgb_ld', ctypes.c_int32),
                ('a_pixelnorm_eps', ctypes.c_float), ('col_stats', ctypes.c_void_p)]


class NoOutput:
    """Stands in for the output tensor of a launch that stores none (wgs_conv_desc.y = NULL with rgb_out): carries the shape only."""

    def __init__(self, *shape):
        self.shape = tuple(shape)

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return None


class WgradDesc(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('dy', ctypes.c_void_p), ('dw', ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ('B', 'Hi', 'Wi', 'Ci', 'Ho', 'Wo', 'Co', 'isy', 'isx', 'ntaps',
                                              'ksplit')] + \
               [('w_tap_stride', ctypes.c_int64), ('w_row_stride', ctypes.c_int64),
                ('dy_t', ctypes.c_int8 * 64), ('dx_t', ctypes.c_int8 * 64), ('wt', ctypes.c_int16 * 64),
                ('precision', ctypes.c_int32), ('x_s2d', ctypes.c_int32), ('ws', ctypes.c_void_p), ('ws_bytes', ctypes.c_int64)]


# Arithmetic of a generator's implicit-GEMM launches (wgs_conv_desc.precision, include/wgs.h):
#   0 'fp32'   exact fp32 MFMA                                   (the reference's arithmetic, 157 TF ceiling)
#   1 'bf16x3' split-bf16, 3 MFMAs per product, ~2^-16           (fp32-class: image error ~1e-5)
#   2 'f16'    fp16 operands, 1 MFMA per product, fp32 accumulate (image error ~8e-4 at 256^2: ON the 1e-3 gate, reported only)
#   3 'f16x2'  fp16 activations x (hi + lo) fp16 weights, 2 MFMAs (image error ~5e-4)
#   4 'mixed'  StyleGAN2 only: per-layer arithmetic from an error budget, see MixedPolicy
#   6 'mixed-strict'  as 'mixed' with a per-layer table CALIBRATED on the generator's own weights so that no single image of a 2 304-image
#              sample is over the 1e-3 gate (STRICT_LADDER; the default table holds the gate per batch tensor and per image at p99 - p99.9)
#   5 'fp32w'  fp32 with the 3x3 stride-1 convs in Winograd F(2x2,3x3) form on the fp32 matrix cores (2.25x fewer multiplies, ~1e-6
#              against the direct form, 3e-6 against fp64: no wider than the direct fp32 kernel); the rest as 'fp32'
#   7 'bf16x3w' split-bf16 (the arithmetic of 'bf16x3': fp32-class) with the 3x3 stride-1 convs' horizontal taps in the Winograd form F(2,3)
#              (csrc/conv_wino_bf16.hip: two bf16 MFMAs per direct product instead of three); launches it does not cover run 'bf16x3'
#  -1 'auto'   per generator: the cheapest mode whose measured image error stays inside the north_star's 1e-3 gate for that
#              architecture with margin (tests/test_precision_schemes_gpu.py): see AUTO_TABLE
# There is NO process-wide arithmetic state: a mode is an attribute of a generator instance (`G.precision`), an argument of
# its forward (`G(z, shift, precision=...)`) and of the step engine (`TrainStep(..., precision=..., r_precision=...)`); bare
# conv calls without `precision=` run the reference's arithmetic (exact fp32).
PRECISION_NAMES = {'auto': -1, 'fp32': 0, 'bf16x3': 1, 'f16': 2, 'f16x2': 3, 'mixed': 4, 'fp32w': 5, 'mixed-strict': 6, 'bf16x3w': 7}
AUTO, MIXED, FP32W, MIXED_STRICT, BF16W = -1, 4, 5, 6, 7
# default of the TRAINING CLIs (train.py, bench.py extra runs); the image-producing CLIs (traverse_latent_space.py,
# sample_gan.py) default to IMAGE_DEFAULT_PRECISION, the fp32-class mode
DEFAULT_PRECISION = 'auto'
IMAGE_DEFAULT_PRECISION = 'bf16x3'
# (generator family, output resolution) -> mode for 'auto'; only entries with a measurement behind them
# (profiles/r4_precision_schemes.json: ProgGAN-256 3.5e-4, ProgGAN-1024 9e-5 per image in f16); everything else falls back to the fp32-class mode.
# StyleGAN2: the CALIBRATED per-layer table (round 6; VERDICT r5 #1: the un-calibrated 'mixed' table sat ON the gate on bench.py's initialisation)
AUTO_TABLE = {('stylegan2', 256): 'mixed-strict', ('stylegan2', 1024): 'mixed-strict', ('proggan', 256): 'f16', ('proggan', 1024): 'f16'}
AUTO_FALLBACK = 'bf16x3'
# ... StyleGAN2 (any size): the same arithmetic with the F(2,3) form of its stride-1 3x3 convs (conv_wino_bf16.hip; what the kernel does not cover is a bf16x3 launch)
AUTO_FALLBACK_BY_FAMILY = {'stylegan2': 'bf16x3w'}


class MixedPolicy:
    """Per-layer arithmetic of a StyleGAN2 generator under 'mixed': every fp16-rounded layer adds an independent ~2.5e-4 (both
    operands rounded, 'f16') or ~1.7e-4 (one operand, 'f16x2') to the image error, and the generator's MACs sit in the layers
    at >= 64x64; so those run in fp16 and the low-resolution layers (latency-bound) in split-bf16.
    table: {output resolution: (stride-1 conv code, up-conv code)}; resolutions not listed run `below` — a code, or a (stride-1, up-conv)
    pair; default (BF16W, 1): fp32-class split-bf16, the stride-1 convs in the F(2,3) form where conv_wino_bf16.hip covers them (maps
    >= 32 x 32 that fill the chip; the rest are direct split-bf16 launches).
    bwd_up_f16: the up-sampling layers' INPUT-GRADIENT convs run in plain f16 when their forward is f16x2 (the two-MFMA form
    exists for the image-error budget; the gradient has its own gate, the shared-gate gradient error).
    bwd_table: explicit arithmetic of the input-gradient convs of the listed resolutions."""

    def __init__(self, table, below=(BF16W, 1), bwd_up_f16=True, bwd_table=None):
        self.table, self.bwd_up_f16 = dict(table), bwd_up_f16
        self.below = tuple(below) if isinstance(below, (tuple, list)) else (below, below)
        self.bwd_table = dict(bwd_table or {})     # {output resolution: (stride-1, up-conv)} arithmetic of the INPUT-GRADIENT convs only

    def fwd(self, out_res, is_up):
        s1, up = self.table.get(out_res, self.below)
        return up if is_up else s1

    def bwd(self, out_res, is_up):
        if out_res in self.bwd_table:
            s1, up = self.bwd_table[out_res]
            return up if is_up else s1
        lp = self.fwd(out_res, is_up)
        return 2 if (is_up and lp == 3 and self.bwd_up_f16) else lp

    def spends(self):
        """Number of layers that round an operand to fp16 in the forward pass (what the image-error budget is spent on)."""
        return sum((s1 in (2, 3)) + (up in (2, 3)) for s1, up in self.table.values())


# Chosen from measured sweeps (tools/policy_sweep.py, profiles/r3_policy_sweep.md): per-image error of every candidate over 384
# (256^2) / 64 (1024^2) latent codes and two weight fills, against the exact-fp32 kernels, and the step time under it.
# StyleGAN2-256: fp16 in the four layers at 128^2 / 256^2 (f16 stride-1 convs, f16x2 up-convs), split-bf16 below.  Every one of the
# 384 images within 1e-3 (max 8.5e-4, p99 7.3e-4, batch tensors <= 5.2e-4).  Round 2's table also ran the two 64^2 layers in fp16
# (10 % faster, but p99 1.0e-3 and 1 % of single images over the gate).  Their INPUT-GRADIENT convs do run in plain fp16 (bwd_table):
# the image gate is a forward matter, the gradient has its own (shared-gate gradient error vs the fp64 oracle, 1e-3).
# Round 6: the split-bf16 stride-1 layers (64^2 and below) take the F(2,3) form (BF16W): same error class, two MFMAs per product instead of three.
W = BF16W
MIXED_256 = MixedPolicy({64: (W, 1), 128: (2, 3), 256: (2, 3)}, bwd_table={64: (2, 2)})
# StyleGAN2-1024 (17 layers): fp16 x2 in the four HBM-bound layers at 512^2 / 1024^2 (64 / 32 channels: the 3-MFMA split-bf16 form
# is what costs there, not the second fp16 MFMA), split-bf16 everywhere else: 12 % faster than split-bf16 everywhere, every
# measured image within 1e-3 (max 8.5e-4).  fp16 in the 64^2 .. 256^2 layers as well measures 1.1e-3 .. 1.4e-3 at this depth.
MIXED_1024 = MixedPolicy({512: (3, 3), 1024: (3, 3)}, bwd_table={64: (2, 2), 128: (2, 2), 256: (2, 2)})
MIXED_POLICIES = {256: MIXED_256, 1024: MIXED_1024}
# 'mixed-strict' (VERDICT r4 #7, r5 #1): NO single image over the 1e-3 gate.  Which table achieves that depends on the weights: the sweep's
# well-conditioned fills (profiles/r5_policy_sweep.md: 4 fills x 576 codes) put the default table at max 1.06e-3 / 0.1 % over, but on
# bench.py's raw constructor initialisation (a mapping network that collapses every z onto nearly one w) the same table measures
# 1.27e-3 / 1.0 %.  So the strict mode is CALIBRATED: a step engine built with 'mixed-strict' — what 'auto' means for StyleGAN2 since
# round 6 — measures the ladder below on its own generator (trainer.TrainStep.calibrate_strict: 2 304 codes against the exact-fp32
# kernels) and takes the first — cheapest — table whose worst single image stays under 0.95 x the gate; the last rung is split-bf16
# everywhere.  Round 6: the rungs keep the stride-1 layers they take out of fp16 in the F(2,3) split-bf16 form (W: fp32-class at 2 MFMAs
# per product), so the budget is spent on the up-sampling layers, where fp16 buys most.  Backward arithmetic is the default table's on
# every rung (the gradient has its own gate, section 3.2 of DESIGN.md).  Order = measured step time (tools/policy_sweep.py, profiles/r6_policy_sweep.md).
_BWD_256 = {64: (2, 2), 128: (2, 2), 256: (2, 2)}
_BWD_1024 = {64: (2, 2), 128: (2, 2), 256: (2, 2), 512: (3, 2), 1024: (3, 2)}
def _p256(table):
    return MixedPolicy(table, bwd_table=_BWD_256)


# 256: ordered by the measured step time on bench.py's generator (profiles/r6_policy_sweep.md, same box: 23.8 / 24.5 / 24.7 / 24.9 / 25.0 / 25.4 / 25.6 /
# 26.0 / 26.4 / 26.7 ms; worst single image of 2 304: 1.28e-3 / 1.19e-3 / 1.09e-3 / 9.5e-4 / 1.06e-3 / 8.2e-4 / 8.1e-4 / 6.6e-4 / 4.7e-4 / 4.4e-5)
STRICT_LADDER = {
    256: [('default 128:2,3;256:2,3', MIXED_256),
          ('128:2,2;256:w,3', _p256({128: (2, 2), 256: (W, 3)})),
          ('128:w,3;256:2,3', _p256({128: (W, 3), 256: (2, 3)})),
          ('128:w,2;256:w,2', _p256({128: (W, 2), 256: (W, 2)})),
          ('128:2,3;256:w,3', _p256({128: (2, 3), 256: (W, 3)})),
          ('128:w,2;256:w,3', _p256({128: (W, 2), 256: (W, 3)})),
          ('128:w,3;256:w,2', _p256({128: (W, 3), 256: (W, 2)})),
          ('128:w,3;256:w,3', _p256({128: (W, 3), 256: (W, 3)})),
          ('128:w,1;256:w,3', _p256({256: (W, 3)})),
          ('bf16x3w everywhere', _p256({}))],
    # 1024 (profiles/r6_policy_sweep.md, bench.py's generator, B = 8): 30.6 / 30.9 / 30.8 ms, worst single image of 576: 1.02e-3 / 4.9e-4 / 4.6e-5 — with the
    # F(2,3) form in the 64^2 .. 256^2 layers the fp16 layers at 512^2 / 1024^2 (HBM-bound) buy nothing any more
    1024: [('default 512:3,3;1024:3,3', MIXED_1024),
           ('512:1,3;1024:1,3', MixedPolicy({512: (1, 3), 1024: (1, 3)}, bwd_table=_BWD_1024)),
           ('bf16x3w everywhere', MixedPolicy({}, bwd_table=_BWD_1024))],
}
STRICT_IMAGES = {256: 2304, 1024: 576}       # latent codes of a calibration (72 batches of the configurations' 32 / 8 images)
# before an engine has calibrated (a bare generator call with precision='mixed-strict'): the most conservative fp16 rung
MIXED_256_STRICT = STRICT_LADDER[256][8][1]
MIXED_STRICT_POLICIES = {256: MIXED_256_STRICT, 1024: STRICT_LADDER[1024][1][1]}


def is_mixed(code):
    return code in (MIXED, MIXED_STRICT)


def mixed_policy(size, code=MIXED):
    if code == MIXED_STRICT:
        return MIXED_STRICT_POLICIES.get(size, MIXED_256_STRICT)
    return MIXED_POLICIES.get(size, MIXED_256)


def layer_precision(code, out_res, is_up, policy=None):
    """Concrete arithmetic of one StyleGAN2 layer's forward conv under mode `code`."""
    if code == FP32W:          # the Winograd form exists for the 3x3 stride-1 convs; the up-convs (1/2/2/4-tap phases) stay direct
        return 0 if is_up else FP32W
    if code == BF16W:          # the F(2,3) form exists for the 3x3 stride-1 convs; the up-convs run the direct split-bf16 kernels
        return 1 if is_up else BF16W
    if not is_mixed(code):
        return code
    return (policy or MIXED_256).fwd(out_res, is_up)


def layer_precision_bwd(code, out_res, is_up, policy=None):
    """Arithmetic of a layer's INPUT-GRADIENT conv under mode `code`."""
    if code == FP32W:
        return 0 if is_up else FP32W
    if code == BF16W:
        return 1 if is_up else BF16W
    if not is_mixed(code):
        return code
    return (policy or MIXED_256).bwd(out_res, is_up)


def precision_code(name):
    if isinstance(name, int) and not isinstance(name, bool):
        if name not in PRECISION_NAMES.values():
            raise L.WgsError("unknown conv precision %r" % (name,))
        return name
    key = str(name).lower()
    if key in PRECISION_NAMES:
        return PRECISION_NAMES[key]
    if key.lstrip('-').isdigit() and int(key) in PRECISION_NAMES.values():
        return int(key)
    raise L.WgsError("unknown conv precision %r (choose from %s)" % (name, ', '.join(PRECISION_NAMES)))


def is_f16_operand(code):
    """Modes that round conv operands to fp16 (codes are labels, not an ordering: 'fp32w' is 5 and is fp32 throughout)."""
    return code in (2, 3, MIXED, MIXED_STRICT)


def is_reduced(code):
    """Modes whose products are not exact fp32 MFMA: the ones the run-time image-error check applies to."""
    return code in (1, 2, 3, MIXED, MIXED_STRICT, BF16W)


def precision_name(code):
    return [k for k, v in PRECISION_NAMES.items() if v == code][0]


def resolve(requested, family, resolution):
    """Concrete precision code (0..4) of generator `family` at `resolution` for the requested mode (name or code; None = 'auto')."""
    code = AUTO if requested is None else precision_code(requested)
    if code != AUTO:
        return code
    return PRECISION_NAMES[AUTO_TABLE.get((family, resolution), AUTO_FALLBACK_BY_FAMILY.get(family, AUTO_FALLBACK))]


# Profiling hook (bench.py): set to a list to collect (shape label, algorithmic FLOPs, start event, end event, kernel symbol) per
# launch; the events are recorded on torch's current stream, which is the stream the kernels are launched on, and the symbol is
# the library's own record of the kernel it dispatched to (wgs_dev_last_kernel, needs wgs_dev_trace_kernels(1)).
PROFILE = None


def _p(t):
    return None if t is None else t.data_ptr()


# Workspace (wgs_conv_desc.ws): one caller-owned buffer per device, reused by every launch on the stream — split-K
# partial tiles (<= 64 MiB) or, for launches with pre-split weights, the split activation planes (4 bytes per input element).
WS_BYTES = 64 << 20
_WS = {}


def _workspace(device, nbytes=0):
    # one buffer per (device, stream): launches on different streams may run concurrently
    key = (device, L.stream_id(device.index))
    ws = _WS.get(key)
    need = max(WS_BYTES, nbytes)
    if ws is None or ws.numel() * 4 < need:
        ws = _WS[key] = torch.empty(need // 4, device=device, dtype=torch.float32)
    return ws


def split_weight(w, precision=1):
    """Pre-split a (frozen) packed fp32 weight into the 16-bit planes of `precision` for the DMA-fed conv kernels:
    bf16 hi / lo (1, wgs_split_bf16), fp16 hi (2) or fp16 hi / lo (3, wgs_split_f16).
    Returns (hi, lo) int16 tensors of w's shape (lo None for precision 2); pass them as w_split= to the conv functions."""
    if not (w.is_cuda and w.is_contiguous() and w.dtype == torch.float32 and w.numel() % 4 == 0):
        raise L.WgsError("split_weight needs a contiguous fp32 GPU tensor with numel % 4 == 0")
    hi = torch.empty_like(w, dtype=torch.int16)
    lo = torch.empty_like(w, dtype=torch.int16) if precision != 2 else None
    if precision == 1:
        L.check(L.lib().wgs_split_bf16(L.ptr(w), L.ptr(hi, torch.int16), L.ptr(lo, torch.int16), ctypes.c_int64(w.numel()), L.stream()), 'wgs_split_bf16')
    elif precision in (2, 3):
        L.check(L.lib().wgs_split_f16(L.ptr(w), L.ptr(hi, torch.int16), L.ptr(lo, torch.int16), ctypes.c_int64(w.numel()), L.stream()), 'wgs_split_f16')
    else:
        raise L.WgsError("split_weight: precision must be 1, 2 or 3")
    return hi, lo


class SplitCache:
    """Lazily built 16-bit planes of one frozen weight tensor, per precision (tests switch the arithmetic at run time)."""

    def __init__(self, w):
        self.w, self.planes = w, {}

    def get(self, precision):
        if precision in (0, FP32W):
            return None
        if precision == BF16W:     # the split-bf16 planes serve the launches the F(2,3) kernel does not cover
            precision = 1
        if precision not in self.planes:
            self.planes[precision] = split_weight(self.w, precision)
        return self.planes[precision]


class WinoCache(SplitCache):
    """Cache of a frozen weight tensor's Winograd U operands only ('fp32w'): no 16-bit planes, so the 16-bit modes keep the kernel
    routing of a launch without pre-split weights (BigGAN / SNGAN convs)."""

    def get(self, precision):
        return None


class StepWinoCache(WinoCache):
    """Winograd U operands of a TRAINED weight tensor, owned by a training engine: the launches that built an operand leave their
    descriptor behind (`recipes`), refresh() rebuilds every operand in place from the tensor's current values — once per optimisation
    step, after the update and off the critical path (reconstructor.StepWeights), instead of one transform launch in front of every
    conv.  Valid only while its owner refreshes it: nothing else may hold one across a weight update.
    A recipe keeps the weight TENSOR it was recorded from (not just the descriptor's raw pointer): refresh() re-reads the pointer from
    it, and a launch whose weights live elsewhere by now (parameters re-homed into another flat bucket, R.to(), a re-allocated
    transposed copy) does not hit the cache — it rebuilds the operand from its own tensor and replaces the recipe."""

    def __init__(self):
        self.w, self.planes, self.recipes = None, {}, {}

    def valid_for(self, key, w):
        r = self.recipes.get(key)
        return key in self.planes and r is not None and r[1].data_ptr() == w.data_ptr() and r[1].device == w.device

    def refresh(self):
        for key, (d, w) in self.recipes.items():
            d.w = w.data_ptr()
            L.check(L.lib().wgs_conv_wino_weight(ctypes.byref(d), L.ptr(self.planes[key]), L.stream()), 'wgs_conv_wino_weight')


def _timed(kind, flops, fn):
    if PROFILE is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    PROFILE.append((kind, flops, s, e, (L.lib().wgs_dev_last_kernel() or b'').decode()))
    return r


def _desc(x, w, y, taps, Hg, Wg, isy=1, osy=1, oy0=0, ox0=0, w_tap_stride=None, w_row_stride=None,
          a_scale=None, col_scale=None, bias=None, noise=None, noise_w=None, act_slope=1.0, gain=1.0,
          a_ld=0, col_ld=0, ups=0, alpha=1.0, addend=None, add_ups=0, act=0, precision=None, into=None, w_split=None,
          a_amax=None, a_bound=1.0, grad_operand=False, a_amax2=None, y_amax=None, x_f16=False, rgb=None, pixelnorm_eps=0.0, col_stats=None):
    """Fill a wgs_conv_desc.  taps: list of (dy, dx, weight_tap_index).  x [B,Hi,Wi,Ci], y [B,Ho,Wo,Co] (NHWC, contiguous).
    x_f16: x is the int16 tensor holding the operand's fp16 plane (wgs_conv_desc.x_f16), written by the producing kernel."""
    if not (x.is_cuda and x.is_contiguous() and y.is_contiguous() and x.dtype == (torch.int16 if x_f16 else torch.float32)):
        raise L.WgsError("conv launch needs contiguous fp32 GPU tensors (no CPU fallback)")
    d = ConvDesc() if into is None else into
    d.x, d.w, d.y = (None if x_f16 else x.data_ptr()), w.data_ptr(), y.data_ptr()
    d.x_f16 = x.data_ptr() if x_f16 else None
    if rgb is not None:       # ToRGB in the epilogue: dict(out=[B,Ho,Wo,4], s=<style rows>, ld=<row stride>, w=[3,Co], scale=)
        d.rgb_out, d.rgb_s, d.rgb_w, d.rgb_scale, d.rgb_ld = rgb['out'].data_ptr(), rgb['s'].data_ptr(), rgb['w'].data_ptr(), rgb['scale'], rgb['ld']
    else:
        d.rgb_out, d.rgb_s, d.rgb_w, d.rgb_scale, d.rgb_ld = None, None, None, 0.0, 0
    d.a_scale, d.col_scale, d.bias = _p(a_scale), _p(col_scale), _p(bias)
    d.noise, d.noise_w = _p(noise), _p(noise_w)
    d.B, d.Hi, d.Wi, d.Ci = x.shape
    d.Hg, d.Wg, d.isy, d.isx = Hg, Wg, isy, isy
    _, d.Ho, d.Wo, d.Co = y.shape
    d.osy, d.osx, d.oy0, d.ox0 = osy, osy, oy0, ox0
    d.ntaps = len(taps)
    d.a_ld, d.col_ld = a_ld, col_ld
    d.ups, d.add_ups, d.act, d.alpha, d.addend = ups, add_ups, act, alpha, _p(addend)
    prec = 0 if precision is None else precision      # a bare conv call: the reference's arithmetic
    if prec == AUTO:           # a bare conv call outside a generator: the fp32-class mode
        prec = PRECISION_NAMES[AUTO_FALLBACK]
    if is_mixed(prec):         # per-layer policies are resolved by the generator (stylegan2.py); elsewhere: fp16 x2
        prec = 3
    if prec == FP32W:          # launch() routes the launches the Winograd kernel covers; everything else is the direct fp32 form
        prec = 0
    if prec == BF16W:          # likewise: what conv_wino_bf16.hip does not cover is a direct split-bf16 launch
        prec = 1
    if grad_operand and prec >= 2 and a_amax is None:
        # an fp16 gradient operand needs a magnitude bound (5 exponent bits); without one the launch runs in split-bf16
        prec = 1
    d.precision = prec
    d.a_amax, d.a_bound, d.a_amax2, d.y_amax = _p(a_amax), a_bound, _p(a_amax2), _p(y_amax)
    d.col_stats = col_stats.data_ptr() if col_stats is not None else None      # BatchNorm scratch (float64): sum y / sum y^2 of the output from the epilogue
    d.a_pixelnorm_eps = pixelnorm_eps       # > 0: the operand is PixelNorm(x), normalised inside the few-channel kernel (pixelnorm_fused_ok)
    if isinstance(w_split, SplitCache):
        w_split = w_split.get(prec)
    d.w_tap_stride, d.w_row_stride = w_tap_stride, w_row_stride
    d.act_slope, d.gain = act_slope, gain
    ws = _workspace(x.device, x.numel() * 4 if w_split is not None else 0)
    d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4
    if w_split is not None:
        d.w_hi, d.w_lo = w_split[0].data_ptr(), _p(w_split[1])
    else:
        d.w_hi, d.w_lo = None, None
    for i, (ty, tx, ti) in enumerate(taps):
        d.dy[i], d.dx[i], d.wt[i] = ty, tx, ti
    return d, 2.0 * d.B * Hg * Wg * d.Co * d.Ci * len(taps)


def _kind(d, nphase):
    """Label of a launch for bench.py's per-shape accounting: arithmetic, channels, input grid, taps / phases, batch."""
    if PROFILE is None:
        return None
    form = 'up-conv x%d phases' % nphase if nphase > 1 else ('%d taps%s' % (d.ntaps, ' stride %d' % d.isy if d.isy > 1 else ''))
    return 'conv %s %d->%d @%dx%d %s B%d' % (precision_name(d.precision), d.Ci, d.Co, d.Hi << d.ups, d.Wi << d.ups, form, d.B)


def _wino_weight(d, w, cache):
    """U = G g G^T of a launch's weights in the Winograd kernel's staging order (wgs_conv_wino_weight); kept in the weight tensor's
    SplitCache when the caller has one (frozen generator weights), rebuilt per launch otherwise (R's trained weights: ~10 us)."""
    # U's fragment order follows the workgroup shape the launch takes (a function of B, H, W, Co): the layout id is part of the key
    key = ('wino', L.lib().wgs_conv_wino_layout(ctypes.byref(d)), d.w_tap_stride, d.w_row_stride, tuple((d.dy[i], d.dx[i], d.wt[i]) for i in range(9)))
    if isinstance(cache, StepWinoCache):
        if cache.valid_for(key, w):
            return cache.planes[key]
    elif isinstance(cache, SplitCache) and key in cache.planes:
        return cache.planes[key]
    U = torch.empty(16 * d.Ci * d.Co, device=w.device, dtype=torch.float32)
    L.check(L.lib().wgs_conv_wino_weight(ctypes.byref(d), L.ptr(U), L.stream()), 'wgs_conv_wino_weight')
    if isinstance(cache, SplitCache):
        cache.planes[key] = U
        if isinstance(cache, StepWinoCache):
            cache.recipes[key] = (ConvDesc.from_buffer_copy(d), w)
    return U


def _wino16_weight(d, w, cache):
    """U = G g of a launch's weights as bf16 hi / lo planes in conv_wino_bf16.hip's B-fragment order (wgs_conv_wino16_weight): kept in the
    weight tensor's SplitCache when the caller has one (frozen generator weights), rebuilt per launch otherwise."""
    key = ('wino16', d.w_tap_stride, d.w_row_stride, tuple((d.dy[i], d.dx[i], d.wt[i]) for i in range(9)))
    if isinstance(cache, SplitCache) and not isinstance(cache, StepWinoCache) and key in cache.planes:
        return cache.planes[key]
    U = torch.empty(24 * d.Ci * d.Co, device=w.device, dtype=torch.int16)
    L.check(L.lib().wgs_conv_wino16_weight(ctypes.byref(d), L.ptr(U, torch.int16), L.stream()), 'wgs_conv_wino16_weight')
    if isinstance(cache, SplitCache) and not isinstance(cache, StepWinoCache):
        cache.planes[key] = U
    return U


def pixelnorm_fused_ok(x, w, y, taps, Hg, Wg, **kw):
    """True when launch(..., pixelnorm_eps=eps) is covered: the conv's operand PixelNorm(x) is formed while the few-channel kernel stages
    its input patch (wgs_conv_desc.a_pixelnorm_eps), so the normalised tensor is never written (ProgGAN's 16- / 32-channel layers)."""
    kw = dict(kw)
    kw.setdefault('pixelnorm_eps', 1e-8)
    d, _ = _desc(x, w, y, taps, Hg, Wg, **kw)
    return bool(L.lib().wgs_conv_pixelnorm_supported(ctypes.byref(d)))


def launch(x, w, y, taps, Hg, Wg, **kw):
    """One implicit-GEMM launch (wgs_conv_igemm), or — precision 'fp32w' / 'bf16x3w' and a shape they cover — the Winograd kernels
    (wgs_conv_wino: F(2x2,3x3) in fp32; wgs_conv_wino16: F(2,3) x direct in split-bf16)."""
    if kw.get('precision') == BF16W:
        # decided before the weight planes are touched: a covered launch reads U only (no bf16 planes of w are built for it)
        kw_ = {k: v for k, v in kw.items() if k != 'w_split'}
        d, flops = _desc(x, w, y, taps, Hg, Wg, **kw_)
        if L.lib().wgs_conv_wino16_supported(ctypes.byref(d)):
            U = _wino16_weight(d, w, kw.get('w_split'))
            kind = _kind(d, 1)
            _timed(kind.replace('conv bf16x3 ', 'conv bf16x3w ', 1) if kind else None, flops, lambda: L.check(L.lib().wgs_conv_wino16(ctypes.byref(d), L.ptr(U, torch.int16), L.stream()), 'wgs_conv_wino16'))
            return y
        kw = dict(kw, precision=1)          # not covered: the direct split-bf16 launch it stands for
    d, flops = _desc(x, w, y, taps, Hg, Wg, **kw)
    if kw.get('precision') == FP32W and L.lib().wgs_conv_wino_supported(ctypes.byref(d)):
        U = _wino_weight(d, w, kw.get('w_split'))
        kind = _kind(d, 1)
        _timed(kind.replace('conv fp32 ', 'conv fp32w ', 1) if kind else None, flops, lambda: L.check(L.lib().wgs_conv_wino(ctypes.byref(d), L.ptr(U), L.stream()), 'wgs_conv_wino'))
        return y
    _timed(_kind(d, 1), flops, lambda: L.check(L.lib().wgs_conv_igemm(ctypes.byref(d), L.stream()), 'wgs_conv_igemm'))
    return y


def launch_multi(x, w, y, phases, **kw):
    """Launches that differ only in (taps, Hg, Wg, oy0, ox0) — the sub-pixel phases of a transposed conv — through
    wgs_conv_igemm_multi (one merged launch where the library supports it).  phases: list of (taps, Hg, Wg, oy0, ox0)."""
    descs = (ConvDesc * len(phases))()
    flops = []
    for i, (taps, Hg, Wg, oy0, ox0) in enumerate(phases):
        flops.append(_desc(x, w, y, taps, Hg, Wg, oy0=oy0, ox0=ox0, into=descs[i], **kw)[1])
    if PROFILE is not None and not L.lib().wgs_conv_igemm_multi_merges(descs, len(phases)):
        # profiling a launch group that the library issues phase by phase (small maps): time every phase on its own, so that each
        # kernel symbol gets its own FLOPs and duration (the launches themselves are the ones wgs_conv_igemm_multi would issue)
        for i in range(len(phases)):
            _timed(_kind(descs[i], 1) + ' (up-conv phase)', flops[i], lambda i=i: L.check(L.lib().wgs_conv_igemm(ctypes.byref(descs[i]), L.stream()), 'wgs_conv_igemm'))
        return y
    _timed(_kind(descs[0], len(phases)), sum(flops), lambda: L.check(L.lib().wgs_conv_igemm_multi(descs, len(phases), L.stream()), 'wgs_conv_igemm_multi'))
    return y


def conv2d(x, w_packed, k, stride=1, pad=0, out=None, **epi):
    """Forward conv. w_packed: [Co, k*k, Ci] memory."""
    B, Hi, Wi, Ci = x.shape
    Co = w_packed.shape[0]
    Ho = (Hi + 2 * pad - k) // stride + 1
    Wo = (Wi + 2 * pad - k) // stride + 1
    y = out if out is not None else torch.empty(B, Ho, Wo, Co, device=x.device, dtype=x.dtype)
    taps = [(ky - pad, kx - pad, ky * k + kx) for ky in range(k) for kx in range(k)]
    return launch(x, w_packed, y, taps, Ho, Wo, isy=stride, w_tap_stride=Ci, w_row_stride=k * k * Ci, **epi)


def conv2d_dgrad(dy, wt_packed, in_hw, k, stride=1, pad=0, **epi):
    """Gradient w.r.t. the conv input. dy [B,Ho,Wo,Co]; wt_packed [k*k, Ci, Co] memory (repack_w_t)."""
    B, Ho, Wo, Co = dy.shape
    Hi, Wi = in_hw
    Ci = wt_packed.shape[1]
    if stride == 1:
        dx = torch.empty(B, Hi, Wi, Ci, device=dy.device, dtype=torch.float32)
        taps = [(pad - ky, pad - kx, ky * k + kx) for ky in range(k) for kx in range(k)]
        return launch(dy, wt_packed, dx, taps, Hi, Wi, w_tap_stride=Ci * Co, w_row_stride=Co, grad_operand=True, **epi)
    if stride != 2:
        raise L.WgsError("conv2d_dgrad: stride must be 1 or 2")
    phases = []
    for py in range(2):
        for px in range(2):
            taps = [((py + pad - ky) // 2, (px + pad - kx) // 2, ky * k + kx)
                    for ky in range(k) if (py + pad - ky) % 2 == 0
                    for kx in range(k) if (px + pad - kx) % 2 == 0]
            phases.append((py, px, taps))
    # pixels never reached by any tap (e.g. 1x1 stride-2) and pixels past the last window get zero
    dx = torch.zeros(B, Hi, Wi, Ci, device=dy.device, dtype=dy.dtype)
    for py, px, taps in phases:
        Hg, Wg = (Hi - py + 1) // 2, (Wi - px + 1) // 2
        if not taps or Hg <= 0 or Wg <= 0:
            continue
        launch(dy, wt_packed, dx, taps, Hg, Wg, osy=2, oy0=py, ox0=px, w_tap_stride=Ci * Co, w_row_stride=Co, grad_operand=True, **epi)
    return dx


def conv_transpose2d_s2(x, w_packed, k=3, out=None, **epi):
    """F.conv_transpose2d(x, W[in,out,k,k], stride=2, padding=0) on NHWC, as 4 sub-pixel phase GEMMs.
    w_packed [Co, k*k, Ci] memory with out[y,x,co] += x[iy,ix,ci]*w[co, ky*k+kx, ci] at y = 2*iy+ky."""
    B, Hi, Wi, Ci = x.shape
    Co = w_packed.shape[0]
    Ho, Wo = 2 * (Hi - 1) + k, 2 * (Wi - 1) + k
    y = out if out is not None else torch.empty(B, Ho, Wo, Co, device=x.device, dtype=x.dtype)
    phases = []
    for py in range(2):
        for px in range(2):
            taps = [((py - ky) // 2, (px - kx) // 2, ky * k + kx)
                    for ky in range(k) if (py - ky) % 2 == 0 for kx in range(k) if (px - kx) % 2 == 0]
            phases.append((taps, (Ho - py + 1) // 2, (Wo - px + 1) // 2, py, px))
    return launch_multi(x, w_packed, y, phases, osy=2, w_tap_stride=Ci, w_row_stride=k * k * Ci, **epi)


class UpconvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ('x', 'w_hi', 'w_lo', 'y', 'a_scale', 'col_scale', 'bias', 'noise', 'noise_w',
                                               'kernel4x4', 'a_amax', 'a_amax2', 'y_amax')] + \
               [('a_bound', ctypes.c_float), ('alpha', ctypes.c_float)] + \
               [(n, ctypes.c_int32) for n in ('B', 'H', 'Ci', 'Co', 'a_ld', 'col_ld', 'precision')] + \
               [(n, ctypes.c_void_p) for n in ('y_f16', 'y_f16_scale', 'y_f16_bound')] + \
               [('y_f16_mul', ctypes.c_float), ('y_f16_add', ctypes.c_float), ('y_f16_ld', ctypes.c_int32)]


# The fused up-sampling layer (conv_upfused.hip) is taken for the fp16 modes from this input size up (below it a 14 x 14-cell
# tile wastes most of its GEMM rows on the image border; tools/bench_upfused.py).
UPCONV_FUSED_MIN_H = {1: 32, 2: 16, 3: 16}      # (split-bf16: from 32 x 32 inputs, where the 8-wave tile fills the chip at B = 32)


def upconv_fused_ok(H, Ci, Co, precision):
    if precision == 1 and H == 64:      # split-bf16 512 -> 256 @64 -> 128: 1.26 ms fused against 1.21 ms unfused (tools/bench_upfused.py, round 5); 32 -> 64: 0.66 / 0.85, 128 -> 256: 1.36 / 1.71
        return False
    return precision in UPCONV_FUSED_MIN_H and H >= UPCONV_FUSED_MIN_H[precision] and Ci % 32 == 0 and (Co % 64 == 0 or Co == 32)


# Forward planes: the fused up-sampling kernel writes the fp16 operand plane of the stride-1 conv that follows it (that conv's style
# vector and power-of-two scale folded in, wgs_upconv_desc.y_f16), and that conv stages the plane as it is through the patch kernel's
# XF16 form — same bits as staging the fp32 tensor, without the multiply / scale / convert work and with half the operand bytes.
FWD_PLANE = True
# ToRGB of StyleGAN2's 128-channel layer at 256^2 in that conv's epilogue (wgs_conv_desc.rgb_out): its 1 GB output is not read back by a
# ToRGB launch, and in the pass that keeps nothing it is not written at all.
RGB_FUSED = True


def rgb_fused_ok(B, H, Ci, Co, lp, has_plane):
    # (the patch kernel's 128 x 128 tile takes launches of >= 200 tiles: B * H * H / 128)
    if not (RGB_FUSED and has_plane and lp == 2 and H >= 16 and (H & (H - 1)) == 0):
        return False
    if Co == 128:
        return Ci <= 128 and B * H * H // 128 >= 200
    return Co == 256 and B * H * H // 256 >= 200            # the LDS-DMA kernel's 256 x 256 tile


def rgb_halo_ok(x, w_packed, lp, **epi):
    """ToRGB may run in the epilogue of this stride-1 3x3 launch WITHOUT an fp16 operand plane: the few-channel kernel (32 / 64 output
    channels on the 512^2 / 1024^2 maps of StyleGAN2-1024) holds all output channels of its pixels (wgs_conv_rgb_supported)."""
    B, H, W, Ci = x.shape
    Co = w_packed.shape[0]
    if not (RGB_FUSED and lp in (1, 2, 3) and Co in (32, 64)):
        return False
    taps = [(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
    d, _ = _desc(x, w_packed, NoOutput(B, H, W, Co), taps, H, W, w_tap_stride=Ci, w_row_stride=9 * Ci, precision=lp, **epi)
    return bool(L.lib().wgs_conv_rgb_supported(ctypes.byref(d)))


def rgb_wino16_ok(x, w_packed, **epi):
    """ToRGB may run in the epilogue of this stride-1 3x3 launch in the F(2,3) split-bf16 form ('bf16x3w'): 128 output channels, i.e. one
    tile of conv_wino_bf16.hip holds every channel of its pixels (StyleGAN2-256's last layer), and the kernel covers the launch."""
    B, H, W, Ci = x.shape
    Co = w_packed.shape[0]
    if not (RGB_FUSED and Co in (128, 256, 512)):        # (Co > 128: one partial sum per 128-channel block, rgb out = [B,H,W,4 * Co / 128])
        return False
    taps = [(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
    epi = {k: v for k, v in epi.items() if k != 'w_split'}
    dummy = torch.empty(1, device=x.device)
    d, _ = _desc(x, w_packed, NoOutput(B, H, W, Co), taps, H, W, w_tap_stride=Ci, w_row_stride=9 * Ci, precision=BF16W,
                 rgb=dict(out=dummy, s=dummy, w=dummy, scale=1.0, ld=Co), **epi)
    return bool(L.lib().wgs_conv_wino16_supported(ctypes.byref(d)))


def fwd_plane_ok(B, Hout, C, Co_next, lp_next):
    """the up-conv's output [B,Hout,Hout,C] may be handed to the next (stride-1, plain fp16) conv as its operand plane"""
    return FWD_PLANE and lp_next == 2 and C % 32 == 0 and Co_next % 128 == 0 and _plane_fits(B, Hout, Hout, C) and \
        (B * Hout * Hout // 256) * (Co_next // 128) >= 200


def upconv_blur_act(x, w_split, blur_kernel, a_scale, a_ld, col_scale, noise, noise_w, bias, precision,
                    a_amax=None, a_amax2=None, y_amax=None, alpha=1.0, plane=None):
    """StyleGAN2's up-sampling StyledConv in one launch (wgs_sg2_upconv_blur_act): modulated conv_transpose2d(stride 2) +
    Blur(pad (1,1)) + noise + bias + leaky-relu*sqrt(2) (models/StyleGAN2/model.py:201-212,231-241,264).
    x [B,H,H,Ci] NHWC; w_split: SplitCache / (hi, lo) fp16 planes of the [Co,9,Ci] weights; returns y [B,2H,2H,Co].
    plane = dict(scale=<next layer's style rows>, ld=<their row stride>, mul=, add=, keep_y=): also write the next conv's fp16 operand
    plane (needs a_amax); returns (y or None, plane int16 [B,2H,2H,Co], bound [1]) instead of y."""
    B, H, W, Ci = x.shape
    if isinstance(w_split, SplitCache):
        w_split = w_split.get(precision)
    Co = w_split[0].shape[0]
    if not (x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and H == W):
        raise L.WgsError("upconv_blur_act needs a contiguous square fp32 GPU tensor (no CPU fallback)")
    y = torch.empty(B, 2 * H, 2 * H, Co, device=x.device) if (plane is None or plane.get('keep_y', True)) else None
    d = UpconvDesc()
    d.x, d.w_hi, d.w_lo, d.y = x.data_ptr(), w_split[0].data_ptr(), _p(w_split[1]), _p(y)
    if plane is not None:
        if a_amax is None:
            raise L.WgsError("upconv_blur_act: a forward plane needs the magnitude scalar of the input (a_amax)")
        yh = torch.empty(B, 2 * H, 2 * H, Co, device=x.device, dtype=torch.int16)
        yb = torch.empty(1, device=x.device)
        d.y_f16, d.y_f16_scale, d.y_f16_bound = yh.data_ptr(), plane['scale'].data_ptr(), yb.data_ptr()
        d.y_f16_mul, d.y_f16_add, d.y_f16_ld = plane['mul'], plane['add'], plane['ld']
    d.a_scale, d.col_scale, d.bias, d.noise, d.noise_w = _p(a_scale), _p(col_scale), _p(bias), _p(noise), _p(noise_w)
    d.kernel4x4, d.a_amax, d.a_amax2, d.y_amax = _p(blur_kernel), _p(a_amax), _p(a_amax2), _p(y_amax)
    d.a_bound, d.alpha = 1.0, alpha
    d.B, d.H, d.Ci, d.Co, d.a_ld, d.col_ld, d.precision = B, H, Ci, Co, a_ld, Co, precision
    kind = ('conv %s %d->%d @%dx%d up-conv + blur fused B%d' % (precision_name(precision), Ci, Co, H, H, B)) if PROFILE is not None else None
    _timed(kind, 2.0 * B * H * H * 9 * Co * Ci,
           lambda: L.check(L.lib().wgs_sg2_upconv_blur_act(ctypes.byref(d), L.stream()), 'wgs_sg2_upconv_blur_act'))
    return y if plane is None else (y, yh, yb)


def conv_transpose2d_s2_dgrad(dy, wt_packed, k=3, **epi):
    """Gradient of conv_transpose2d_s2 w.r.t. its input = stride-2 conv over dy. wt_packed [k*k, Ci, Co]."""
    B, Ho, Wo, Co = dy.shape
    Ci = wt_packed.shape[1]
    Hi, Wi = (Ho - k) // 2 + 1, (Wo - k) // 2 + 1
    dx = torch.empty(B, Hi, Wi, Ci, device=dy.device, dtype=torch.float32)
    taps = [(ky, kx, ky * k + kx) for ky in range(k) for kx in range(k)]
    return launch(dy, wt_packed, dx, taps, Hi, Wi, isy=2, w_tap_stride=Ci * Co, w_row_stride=Co, grad_operand=True, **epi)


# The transposed blur in front of an up-sampling layer's gradient conv can store its result as that conv's fp16 operand plane
# (wgs_sg2_blur_bwd_f16 -> wgs_conv_desc.x_f16): the (2H+1)^2 fp32 tensor is never written and the conv runs the LDS-DMA kernel
# without a pre-pass, bit-identical to the fp32 route.  Taken for plain-fp16 gradient launches that fill the chip with 256-row tiles.
BLUR_BWD_F16 = True
# The same for the stride-1 layers: sg2_act_bwd stores dy only as the fp16 plane of the gradient conv (wgs_sg2_act_bwd_f16), scaled
# from an a-priori magnitude bound (wgs_sg2_dy_bound) since its own maximum is not known before it has run.
DY_PLANE = True
# Round 6, measured and left OFF: an UP-sampling layer's dy as an fp16 plane as well, where its transposed blur writes the gradient conv's plane
# anyway (blur_bwd_f16_ok): sg2_act_bwd stores 2 instead of 4 bytes per element and the blur reads 2 instead of 4 (wgs_sg2_blur_bwd_f16_x16:
# bit for bit the plane the fp32-input blur makes of the same values; the generator's gradient moves by 5e-5, one more fp16 rounding in front
# of a 16-tap average).  Same-box A/B of the auto step, three alternations: 25.32 - 25.42 ms with the route, 25.13 - 25.37 ms without — the
# 1.9 GB it removes were served from the memory-side cache (the blur reads what sg2_act_bwd has just written), and the route adds a bound launch
# per layer.  tests/test_blur_bwd_f16_gpu.py keeps both forms covered.
DY_PLANE_UP = False
DY_PLANE_MIN_CO = 128       # (round 3: 256 — the plane then had only the LDS-DMA kernel, which re-reads every activation row nine times from L2;
                            #  planes of stride-1 launches with < 512 output columns now go through the patch kernel's XF16 form)
# The LDS-DMA kernel addresses an fp16 operand plane through one buffer descriptor: its extent (2 bytes per element) must stay
# below 2^31 bytes; larger planes (StyleGAN2-256 at a per-GPU batch >= 64: dt [64,257,257,128]) take the fp32 route.
PLANE_MAX_BYTES = (1 << 31) - 1


def _plane_fits(B, H, W, C):
    return B * H * W * C * 2 <= PLANE_MAX_BYTES


def blur_bwd_f16_ok(B, Hc, Ci_dgrad, Co_dgrad, precision):
    """dy [B,Hc,Hc,Ci_dgrad] -> gradient conv to [B,Hc/2,Hc/2,Co_dgrad]"""
    if not BLUR_BWD_F16 or precision != 2 or Ci_dgrad % 32 or Co_dgrad % 128 or not _plane_fits(B, Hc + 1, Hc + 1, Ci_dgrad):
        return False
    return (B * (Hc // 2) ** 2 // 256) * (Co_dgrad // 128) >= 256


def dy_plane_ok(B, Hc, Ci_dgrad, Co_dgrad, precision):
    """sg2_act_bwd may store dy [B,Hc,Hc,Ci_dgrad] only as the fp16 plane of the stride-1 gradient conv to Co_dgrad channels"""
    if not DY_PLANE or precision != 2 or Ci_dgrad % 32 or Co_dgrad % 128 or Co_dgrad < DY_PLANE_MIN_CO or not _plane_fits(B, Hc, Hc, Ci_dgrad):
        return False
    return (B * Hc * Hc // 256) * (Co_dgrad // 128) >= 256


def blur_bwd_f16(dy, blur_f, a_amax, a_bound):
    """dt = transposed blur of dy as the fp16 operand plane of the stride-2 gradient conv; dy fp32, or (int16) the fp16 plane
    sg2_act_bwd_f16 wrote with dy_bound = a_amax (wgs_sg2_blur_bwd_f16_x16)"""
    B, H, W, Cc = dy.shape
    dt = torch.empty(B, H + 1, W + 1, Cc, device=dy.device, dtype=torch.int16)
    if dy.dtype == torch.int16:
        L.check(L.lib().wgs_sg2_blur_bwd_f16_x16(L.ptr(dy, torch.int16), L.ptr(blur_f), L.ptr(dt, torch.int16), L.rawptr(a_amax), L.c_float(a_bound),
                                                 B, H, W, Cc, L.stream()), 'wgs_sg2_blur_bwd_f16_x16')
    else:
        L.check(L.lib().wgs_sg2_blur_bwd_f16(L.ptr(dy), L.ptr(blur_f), L.ptr(dt, torch.int16), L.rawptr(a_amax), L.c_float(a_bound),
                                             B, H, W, Cc, L.stream()), 'wgs_sg2_blur_bwd_f16')
    return dt


WGRAD_DIRECT = True      # stride-1 3x3 weight gradients through conv_wgrad_direct.hip (no LDS staging, no atomics) where the shape allows


def conv2d_wgrad(x, dy, dw_packed, k, stride=1, pad=0, ksplit=0, precision=0, x_s2d=False):
    """Accumulate the weight gradient into the zero-initialised dw_packed [Co, k*k, Ci] memory.
    precision 0: exact fp32 MFMA; 1: split-bf16 x3 (fp32-class) where the shape allows it.
    x_s2d: x is the space-to-depth tensor [B, Hi/2, Wi/2, 32] of a logical [B, Hi, Wi, 8] input (wgs_pack_pair_s2d)."""
    if x_s2d:
        B, Hi, Wi, Ci = x.shape[0], 2 * x.shape[1], 2 * x.shape[2], 8
    else:
        B, Hi, Wi, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    d = WgradDesc()
    d.x, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), dw_packed.data_ptr()
    d.B, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co = B, Hi, Wi, Ci, Ho, Wo, Co
    d.isy, d.isx, d.ntaps, d.ksplit = stride, stride, k * k, ksplit
    d.precision = precision
    d.x_s2d = int(bool(x_s2d))
    if WGRAD_DIRECT and k == 3 and stride == 1 and not x_s2d:
        ws = _workspace(x.device)          # partial tiles of the direct-fragment kernel's pixel-range splits (private to the stream)
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4
    d.w_tap_stride, d.w_row_stride = Ci, k * k * Ci
    i = 0
    for ky in range(k):
        for kx in range(k):
            d.dy_t[i], d.dx_t[i], d.wt[i] = ky - pad, kx - pad, i
            i += 1
    flops = 2.0 * B * Ho * Wo * Co * Ci * k * k
    _timed('wgrad %s %d->%d @%dx%d %d taps B%d' % ('bf16x3' if precision == 1 and Co % 64 == 0 and (Ci % 64 == 0 or (Ci <= 32 and k * k * Ci >= 64 and Wo % 8 == 0)) else 'fp32', Ci, Co, Hi, Wi, k * k, B) if PROFILE is not None else None, flops, lambda: L.check(L.lib().wgs_conv_wgrad(ctypes.byref(d), L.stream()), 'wgs_conv_wgrad'))
    return dw_packed


def repack_w_t(w_packed, Co, T, Ci, out=None):
    """[Co,T,Ci] -> [T,Ci,Co] (weights for dgrad contractions)."""
    dst = out if out is not None else torch.empty(T, Ci, Co, device=w_packed.device, dtype=w_packed.dtype)
    L.check(L.lib().wgs_repack_w_t(L.rawptr(w_packed), L.rawptr(dst), Co, T, Ci, L.stream()), 'wgs_repack_w_t')
    return dst


def pack_weight(w):
    """PyTorch [Co,Ci,kh,kw] -> contiguous [Co, kh*kw, Ci] (host-side, one-off for frozen weights)."""
    Co, Ci, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(Co, kh * kw, Ci).contiguous()
