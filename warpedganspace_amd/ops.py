"""Operator-level mirror of the reference's native seam `models/StyleGAN2/op` (op/__init__.py:1-2):
`FusedLeakyReLU`, `fused_leaky_relu`, `upfirdn2d` — same names, arguments and autograd behaviour,
backed by the C ABI (wgs_bias_act / wgs_upfirdn2d) instead of the JIT-built CUDA extension.
"""
import torch
from torch import nn

from . import _lib as L


def fused_bias_act(x, bias=None, ref=None, act=3, grad=0, alpha=0.2, scale=2 ** 0.5):
    """fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale) -> Tensor
    (models/StyleGAN2/op/fused_bias_act.cpp:11-21): bias indexed on dim 1, output allocated here."""
    x = x.contiguous()
    y = torch.empty_like(x)
    step_b = 1
    for s in x.shape[2:]:
        step_b *= s
    has_b = bias is not None and bias.numel() > 0
    has_r = ref is not None and ref.numel() > 0
    # hold every (possibly re-laid-out) operand in a local until the launch has been issued
    bias_c = bias.contiguous() if has_b else None
    ref_c = ref.contiguous() if has_r else None
    L.check(L.lib().wgs_bias_act(L.ptr(x), L.ptr(bias_c), L.ptr(ref_c), L.ptr(y), act, grad,
                                 L.c_float(alpha), L.c_float(scale), L.c_int64(x.numel()), step_b,
                                 bias.numel() if has_b else 1, L.stream()), 'wgs_bias_act')
    return y


class _FusedLeakyReLU(torch.autograd.Function):
    """FusedLeakyReLUFunction (op/fused_act.py:51-70): backward gated on the saved OUTPUT."""

    @staticmethod
    def forward(ctx, x, bias, negative_slope, scale):
        out = fused_bias_act(x, bias, None, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.cfg = (negative_slope, scale)
        return out

    @staticmethod
    def backward(ctx, gout):
        out, = ctx.saved_tensors
        negative_slope, scale = ctx.cfg
        gin = fused_bias_act(gout, None, out, 3, 1, negative_slope, scale)
        dims = [0] + list(range(2, gin.ndim))
        return gin, gin.sum(dims), None, None


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    return _FusedLeakyReLU.apply(x, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, x):
        return fused_leaky_relu(x, self.bias, self.negative_slope, self.scale)


def upfirdn2d_mhwc(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """upfirdn2d_op.upfirdn2d(input[major,h,w,minor], kernel, ...) (op/upfirdn2d.cpp:12-23)."""
    x = x.contiguous()
    major, in_h, in_w, minor = x.shape
    kh, kw = kernel.shape
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    y = torch.empty(major, out_h, out_w, minor, dtype=x.dtype, device=x.device)
    kernel_c = kernel.contiguous()
    L.check(L.lib().wgs_upfirdn2d(L.ptr(x), L.ptr(kernel_c), L.ptr(y), major, in_h, in_w, minor, kh, kw,
                                  up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, L.stream()),
            'wgs_upfirdn2d')
    return y


class _UpFirDn2d(torch.autograd.Function):
    """UpFirDn2d (op/upfirdn2d.py:87-141): backward = the same op with up<->down swapped, the flipped
    kernel and the g_pad of :110-115."""

    @staticmethod
    def forward(ctx, x, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        px0, px1, py0, py1 = pad
        kh, kw = kernel.shape
        b, c, in_h, in_w = x.shape
        out = upfirdn2d_mhwc(x.reshape(-1, in_h, in_w, 1), kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1)
        out_h, out_w = out.shape[1], out.shape[2]
        ctx.save_for_backward(torch.flip(kernel, [0, 1]))
        ctx.cfg = (up, down, (b, c, in_h, in_w), (out_h, out_w),
                   (kw - px0 - 1, in_w * up_x - out_w * down_x + px0 - up_x + 1,
                    kh - py0 - 1, in_h * up_y - out_h * down_y + py0 - up_y + 1))
        return out.view(b, c, out_h, out_w)

    @staticmethod
    def backward(ctx, gout):
        gkernel, = ctx.saved_tensors
        (up_x, up_y), (down_x, down_y), in_size, (out_h, out_w), (gx0, gx1, gy0, gy1) = ctx.cfg
        gin = upfirdn2d_mhwc(gout.reshape(-1, out_h, out_w, 1), gkernel, down_x, down_y, up_x, up_y, gx0, gx1, gy0, gy1)
        return gin.view(in_size), None, None, None, None


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    return _UpFirDn2d.apply(x, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
