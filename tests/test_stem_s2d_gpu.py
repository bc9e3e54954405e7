"""GPU: the Reconstructor's stem (torchvision resnet18 conv1: 7 x 7, stride 2, pad 3 on cat(x1, x2) — lib/reconstructor.py:54-63,73-78) in
SPACE-TO-DEPTH form: wgs_pack_pair_s2d / wgs_stem_weight_s2d / wgs_unpack_pair_s2d_grad, the 4 x 4-window launches through the
few-channel halo kernel, and the weight gradient reading the s2d input in place (wgs_wgrad_desc.x_s2d).

  * the three re-indexing kernels against their index definitions (exact);
  * conv over the s2d tensor == F.conv2d(cat(x1, x2), w, stride 2, pad 3) in float64 (to the split-bf16 tolerance), forward, input
    gradient and weight gradient; the halo kernel really runs (by symbol) at a size that fills the chip;
  * a whole Reconstructor forward / backward with the stem in the s2d form against the gather form (same arithmetic class)."""
import os

import pytest
import torch
import torch.nn.functional as F

from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C
from warpedganspace_amd import reconstructor as RR

pytestmark = pytest.mark.gpu


def _s2d_ref(x):
    """[B, C8, H, W] -> [B, H/2, W/2, 32] with channel (py*2 + px)*8 + j"""
    B, Cn, H, W = x.shape
    return x.view(B, Cn, H // 2, 2, W // 2, 2).permute(0, 2, 4, 3, 5, 1).reshape(B, H // 2, W // 2, 4 * Cn).contiguous()


def test_pack_unpack_and_weight_maps(dev):
    torch.manual_seed(0)
    lib, st = L.lib(), L.stream()
    B, c, H, W = 3, 3, 16, 24
    x1, x2 = torch.randn(B, c, H, W, device=dev), torch.randn(B, c, H, W, device=dev)
    xs = torch.empty(B, H // 2, W // 2, 32, device=dev)
    L.check(lib.wgs_pack_pair_s2d(L.ptr(x1), L.ptr(x2), L.ptr(xs), B, c, H, W, st))
    x8 = torch.cat([x1, x2, torch.zeros(B, 2, H, W, device=dev)], 1)
    assert torch.equal(xs, _s2d_ref(x8))
    g = torch.randn(B, H // 2, W // 2, 32, device=dev)
    d1, d2 = torch.empty_like(x1), torch.empty_like(x2)
    L.check(lib.wgs_unpack_pair_s2d_grad(L.ptr(g), L.ptr(d1), L.ptr(d2), B, c, H, W, st))
    full = g.view(B, H // 2, W // 2, 2, 2, 8).permute(0, 5, 1, 3, 2, 4).reshape(B, 8, H, W)
    assert torch.equal(d1, full[:, :c]) and torch.equal(d2, full[:, c:2 * c])
    w = torch.randn(64, 49, 2 * c, device=dev)
    wsd = torch.empty(64, 16, 32, device=dev)
    L.check(lib.wgs_stem_weight_s2d(L.ptr(w), L.ptr(wsd), 64, 2 * c, 0, st))
    ref = torch.zeros(64, 16, 32, device=dev)
    for r in range(4):
        for s_ in range(4):
            for py in range(2):
                for px in range(2):
                    ky, kx = 2 * r + py - 1, 2 * s_ + px - 1
                    if 0 <= ky < 7 and 0 <= kx < 7:
                        ref[:, r * 4 + s_, (py * 2 + px) * 8:(py * 2 + px) * 8 + 2 * c] = w[:, ky * 7 + kx]
    assert torch.equal(wsd, ref) and int((wsd != 0).sum()) == 64 * 49 * 2 * c
    back = torch.empty_like(w)
    L.check(lib.wgs_stem_weight_s2d(L.ptr(wsd), L.ptr(back), 64, 2 * c, 1, st))
    assert torch.equal(back, w)


@pytest.mark.parametrize('prec', [1, 0])
@pytest.mark.parametrize('B,H,W,expect_halo', [(2, 32, 64, False), (4, 64, 64, False), (32, 128, 128, True)])
def test_s2d_conv_equals_the_strided_7x7(dev, B, H, W, expect_halo, prec):
    torch.manual_seed(B + H)
    lib, st = L.lib(), L.stream()
    c = 3
    x1, x2 = torch.randn(B, c, H, W, device=dev), torch.randn(B, c, H, W, device=dev)
    w = torch.randn(64, 2 * c, 7, 7, device=dev) / (2 * c * 49) ** 0.5
    wp = C.pack_weight(w)                                               # [64, 49, 6]
    xs = torch.empty(B, H // 2, W // 2, 32, device=dev)
    L.check(lib.wgs_pack_pair_s2d(L.ptr(x1), L.ptr(x2), L.ptr(xs), B, c, H, W, st))
    wsd = torch.empty(64, 16, 32, device=dev)
    L.check(lib.wgs_stem_weight_s2d(L.ptr(wp), L.ptr(wsd), 64, 2 * c, 0, st))
    y = torch.empty(B, H // 2, W // 2, 64, device=dev)
    lib.wgs_dev_trace_kernels(1)
    try:
        C.launch(xs, wsd, y, RR._S2D_TAPS, H // 2, W // 2, w_tap_stride=32, w_row_stride=512, precision=prec)
        sym = lib.wgs_dev_last_kernel().decode()
    finally:
        lib.wgs_dev_trace_kernels(0)
    assert sym.startswith('halo3x3_kernel<0, 32, 64, 4>') == (expect_halo and prec == 1), sym
    tol = 3e-5 if prec == 1 else 3e-6
    xd = torch.cat([x1, x2], 1).double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    ref = F.conv2d(xd, wd, stride=2, padding=3)
    assert (y.double().permute(0, 3, 1, 2) - ref).abs().max() <= tol * ref.abs().max()
    # input gradient: transposed window over dy, then depth-to-space
    gy = torch.randn_like(y)
    ref.backward(gy.double().permute(0, 3, 1, 2))
    wst = C.repack_w_t(wsd, 64, 16, 32)
    dxs = torch.empty(B, H // 2, W // 2, 32, device=dev)
    C.launch(gy, wst, dxs, [(-a, -b, t) for a, b, t in RR._S2D_TAPS], H // 2, W // 2, w_tap_stride=32 * 64, w_row_stride=64, precision=prec, grad_operand=True)
    d1, d2 = torch.empty_like(x1), torch.empty_like(x2)
    L.check(lib.wgs_unpack_pair_s2d_grad(L.ptr(dxs), L.ptr(d1), L.ptr(d2), B, c, H, W, st))
    got = torch.cat([d1, d2], 1).double()
    assert (got - xd.grad).abs().max() <= tol * xd.grad.abs().max()
    # weight gradient: the exact fp32 kernel reading the s2d tensor in place
    dw8 = torch.zeros(64, 49, 8, device=dev)
    C.conv2d_wgrad(xs, gy, dw8, 7, stride=2, pad=3, x_s2d=True)
    refw = wd.grad.permute(0, 2, 3, 1).reshape(64, 49, 2 * c)
    assert (dw8[:, :, :2 * c].double() - refw).abs().max() <= 2e-5 * refw.abs().max()
    assert float(dw8[:, :, 2 * c:].abs().max()) == 0.0
    # ... and the split-bf16 form of it (conv_wgrad16.hip, the 392 (tap, channel) pairs flattened into the GEMM columns): fp32-class
    dw8b = torch.zeros(64, 49, 8, device=dev)
    lib.wgs_dev_trace_kernels(1)
    try:
        C.conv2d_wgrad(xs, gy, dw8b, 7, stride=2, pad=3, x_s2d=True, precision=1)
        symw = lib.wgs_dev_last_kernel().decode()
    finally:
        lib.wgs_dev_trace_kernels(0)
    assert symw.startswith('igemm_wgrad16_kernel<64, 128, true>') == ((W // 2) % 8 == 0), symw
    assert (dw8b[:, :, :2 * c].double() - refw).abs().max() <= 4e-5 * refw.abs().max()
    assert float(dw8b[:, :, 2 * c:].abs().max()) == 0.0
    # the gradient taken in the s2d form (4 x 4 window, 32 channels: what the Reconstructor does) and gathered back to 7 x 7 x 2c
    for p_ in (0, 1):
        dws = torch.zeros(64, 16, 32, device=dev)
        C.conv2d_wgrad(xs, gy, dws, 4, stride=1, pad=2, precision=p_)
        back = torch.empty(64, 49, 2 * c, device=dev)
        L.check(lib.wgs_stem_weight_s2d(L.ptr(dws), L.ptr(back), 64, 2 * c, 1, st))
        assert (back.double() - refw).abs().max() <= (4e-5 if p_ else 2e-5) * refw.abs().max()
    # the same on the plain [B, H, W, 8] input (the gather form of the stem)
    xg = torch.zeros(B, H, W, 8, device=dev)
    xg[..., :2 * c] = torch.cat([x1, x2], 1).permute(0, 2, 3, 1)
    dw8c = torch.zeros(64, 49, 8, device=dev)
    C.conv2d_wgrad(xg, gy, dw8c, 7, stride=2, pad=3, precision=1)
    assert (dw8c[:, :, :2 * c].double() - refw).abs().max() <= 4e-5 * refw.abs().max()


@pytest.mark.parametrize('arith', ['bf16x3', 'fp32', 'fp32w'])
def test_reconstructor_with_the_s2d_stem_vs_the_gather_stem(dev, monkeypatch, arith):
    torch.manual_seed(3)
    B, K, S = 32, 16, 128
    x1, x2 = torch.randn(B, 3, S, S, device=dev), torch.randn(B, 3, S, S, device=dev)
    res = []
    for on in (True, False):
        monkeypatch.setattr(RR, 'STEM_S2D', on)
        torch.manual_seed(5)
        R = RR.Reconstructor('ResNet', K).to(dev).train()
        R.arith = {'bf16x3': RR.R_FP32_CLASS, 'fp32': RR.R_EXACT, 'fp32w': RR.R_FP32_WINO}[arith]
        x2g = x2.clone().requires_grad_(True)
        lg, mg = R(x1, x2g)
        ((lg * torch.linspace(-1, 1, lg.numel(), device=dev).view_as(lg)).sum() + mg.sum()).backward()
        res.append((lg.detach().clone(), mg.detach().clone(), x2g.grad.clone(), R.features_extractor.conv1.weight.grad.clone(),
                    R.features_extractor.layer1[0].conv1.weight.grad.clone()))
    (l1, m1, g1, w1, v1), (l0, m0, g0, w0, v0) = res
    tol = 2e-4 if arith == 'bf16x3' else 2e-5
    assert (l1 - l0).abs().max() <= tol * l0.abs().max() and (m1 - m0).abs().max() <= tol * m0.abs().max()
    cos = lambda a, b: float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()))
    print(arith, 's2d vs gather stem: d_img cosine %.6f, conv1 dW cosine %.6f, layer1 dW cosine %.6f' % (cos(g1, g0), cos(w1, w0), cos(v1, v0)))
    # two evaluations of the same network in the same arithmetic class: a few ReLU gates / max-pool winners differ, the direction holds
    lim = 0.999 if arith == 'bf16x3' else 0.9999
    assert cos(g1, g0) > lim and cos(w1, w0) > lim and cos(v1, v0) > lim
