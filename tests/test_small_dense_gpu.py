"""GPU: the small dense kernels of the StyleGAN2 path — EqualLinear forward (8 batch rows per trip, eight-value wave reduction) against
float64, and the batched style-gradient reduction (wgs_sg2_style_grad_batch) against one wgs_sg2_style_grad launch per layer."""
import ctypes

import pytest
import torch

from tests.util import rel_err
from warpedganspace_amd import _lib as L
from warpedganspace_amd.stylegan2 import StyleGradBatch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('M,N,K,epi', [(32, 512, 512, 1), (5, 100, 256, 0), (17, 8704, 512, 0), (64, 64, 1024, 1), (3, 7, 2048, 2)])
def test_linear_fwd_vs_float64(dev, M, N, K, epi):
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K)
    w = torch.randn(N, K) / K ** 0.5
    b = torch.randn(N)
    y = torch.empty(M, N, device=dev)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)        # (kept alive: L.ptr() of a temporary would dangle before the launch)
    L.check(L.lib().wgs_linear_fwd(L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(y), M, N, K, K, N, L.c_float(0.7),
                                   L.c_float(0.3), 1 if epi == 2 else 0, epi, L.c_float(1e-3), L.c_float(1.5), L.stream()), 'linear_fwd')
    xin = x.double() ** 2 if epi == 2 else x.double()
    v = 0.7 * xin @ w.double().t() + 0.3 * b.double()
    if epi == 1:
        v = torch.nn.functional.leaky_relu(v, 0.2) * 2 ** 0.5
    elif epi == 2:
        v = torch.rsqrt(v.clamp_min(0) + 1e-3) if (v + 1e-3 > 0).all() else torch.rsqrt(v + 1e-3)
    ref = v * 1.5
    ok = torch.isfinite(ref)
    assert rel_err(y.cpu()[ok], ref[ok]) < 2e-6


def test_style_grad_batch_vs_single_launches(dev):
    torch.manual_seed(3)
    B, sumC = 4, 64 + 128 + 32
    layers = [(64, 128, True), (128, 64, True), (3, 32, False)]          # (Co, Ci, demodulated)
    S = torch.randn(B, sumC, device=dev)
    dS0, dS1 = torch.zeros(B, sumC, device=dev), torch.zeros(B, sumC, device=dev)
    lib, st = L.lib(), L.stream()
    sb = StyleGradBatch()
    sb.n, sb.B, sb.ld_s, sb.ld_out = len(layers), B, sumC, sumC
    keep, off = [], 0
    for k, (Co, Ci, dm) in enumerate(layers):
        num = torch.randn(B, Co, device=dev) if dm else None
        demod = (torch.rand(B, Co, device=dev) + 0.5) if dm else None
        wsq = torch.rand(Co, Ci, device=dev) if dm else None
        dsdir = torch.randn(B, Ci, device=dev)
        keep.append((num, demod, wsq, dsdir))
        L.check(lib.wgs_sg2_style_grad(L.ptr(num), L.ptr(demod), L.rawptr(S[:, off:]), L.ptr(dsdir), L.ptr(wsq), L.c_float(1.0),
                                       L.rawptr(dS0[:, off:]), B, Co, Ci, sumC, sumC, st), 'style_grad')
        sb.num[k] = None if num is None else num.data_ptr()
        sb.demod[k] = None if demod is None else demod.data_ptr()
        sb.wsq[k] = None if wsq is None else wsq.data_ptr()
        sb.s[k], sb.dsdir[k], sb.dstyle[k] = S[:, off:].data_ptr(), dsdir.data_ptr(), dS1[:, off:].data_ptr()
        sb.Co[k], sb.Ci[k], sb.scale2[k] = Co, Ci, 1.0
        off += Ci
    L.check(lib.wgs_sg2_style_grad_batch(ctypes.byref(sb), st), 'style_grad_batch')
    assert torch.equal(dS0, dS1)
    assert dS0.abs().sum() > 0
