"""GPU: the direct-fragment weight-gradient kernel (csrc/conv_wgrad_direct.hip, round 5) — the weight gradient of torchvision's BasicBlock
3x3 stride-1 convs (lib/reconstructor.py:52-79 -> torchvision resnet18; restated at oracle/wgs_oracle.py:296-319) with MFMA operand
fragments loaded straight from global memory, partial tiles per pixel-range split and a split-ordered reduction: against float64
autograd, against the LDS-staged kernels it replaces, bit-reproducible, accumulating into dw."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_err
from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _case(B, Ci, Co, H, W, seed):
    torch.manual_seed(seed)
    x = torch.randn(B, Ci, H, W, dtype=torch.float64)
    w = (torch.randn(Co, Ci, 3, 3, dtype=torch.float64) / (Ci * 9) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, padding=1)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    return x, g, w.grad.permute(0, 2, 3, 1).reshape(Co, 9, Ci)


def _run(xd, gd, Co, Ci, precision, ksplit=0, into=None):
    dw = torch.zeros(Co, 9, Ci, device=xd.device) if into is None else into
    L.lib().wgs_dev_trace_kernels(1)
    C.conv2d_wgrad(xd, gd, dw, 3, stride=1, pad=1, precision=precision, ksplit=ksplit)
    sym = L.lib().wgs_dev_last_kernel().decode()
    L.lib().wgs_dev_trace_kernels(0)
    return dw, sym


@pytest.mark.parametrize('precision,tol', [(1, 3e-5), (0, 6e-6)])      # (one split = one fp32 chain over up to 131 072 pixels per wave)
@pytest.mark.parametrize('B,Ci,Co,H,W', [(4, 64, 64, 16, 16), (2, 128, 64, 24, 8), (3, 64, 128, 8, 32), (32, 64, 64, 64, 64), (2, 512, 512, 8, 8),
                                         (1, 256, 128, 16, 16), (5, 64, 192, 16, 40)])
def test_direct_wgrad_vs_float64(dev, precision, tol, B, Ci, Co, H, W):
    x, g, ref = _case(B, Ci, Co, H, W, Ci + Co + H + W)
    xd, gd = nhwc(x.float()).to(dev), nhwc(g.float()).to(dev)
    for ksplit in (0, 1, 5):
        dw, sym = _run(xd, gd, Co, Ci, precision, ksplit)
        assert sym == 'wgrad_direct_kernel<%d>' % (0 if precision == 1 else 4), sym
        e = rel_err(dw, ref)
        assert e < tol, (ksplit, e)
    # bit-reproducible: no atomics anywhere (the staged kernels combine their K splits with atomicAdd)
    again, _ = _run(xd, gd, Co, Ci, precision, 5)
    assert torch.equal(dw, again)
    # accumulates into dw (the trainer's flat gradient bucket is zeroed once per step)
    pre = torch.randn(Co, 9, Ci, device=dev)
    acc, _ = _run(xd, gd, Co, Ci, precision, 5, into=pre.clone())
    assert rel_err(acc - pre, dw) < 1e-5


def test_direct_wgrad_agrees_with_the_staged_kernels_and_declines_what_it_does_not_cover(dev, monkeypatch):
    B, Ci, Co, H = 4, 128, 128, 16
    x, g, ref = _case(B, Ci, Co, H, H, 77)
    xd, gd = nhwc(x.float()).to(dev), nhwc(g.float()).to(dev)
    d1, s1 = _run(xd, gd, Co, Ci, 1)
    d0, s0 = _run(xd, gd, Co, Ci, 0)
    monkeypatch.setenv('WGS_WGRAD_STAGED', '1')
    L.lib().wgs_dev_reload_flags()
    try:
        o1, t1 = _run(xd, gd, Co, Ci, 1)
        o0, t0 = _run(xd, gd, Co, Ci, 0)
    finally:
        monkeypatch.delenv('WGS_WGRAD_STAGED')
        L.lib().wgs_dev_reload_flags()
    assert s1.startswith('wgrad_direct_kernel<0') and s0.startswith('wgrad_direct_kernel<4')
    assert t1.startswith('igemm_wgrad16') and t0.startswith('igemm_wgrad_kernel'), (t1, t0)
    assert rel_err(d1, o1) < 3e-5 and rel_err(d0, o0) < 3e-6
    # not covered: width % 8 != 0, pixel count % 16 != 0, strided, 1x1 — the staged kernels take those launches
    for (b, h, w_) in ((2, 9, 9), (1, 3, 8)):
        xx, gg, rr = _case(b, 64, 64, h, w_, 5)
        dd, sym = _run(nhwc(xx.float()).to(dev), nhwc(gg.float()).to(dev), 64, 64, 1)
        assert not sym.startswith('wgrad_direct'), sym
        assert rel_err(dd, rr) < 5e-5
    # a caller without a workspace (the C ABI's older clients): staged kernels
    monkeypatch.setattr(C, 'WGRAD_DIRECT', False)
    dd, sym = _run(xd, gd, Co, Ci, 1)
    assert sym.startswith('igemm_wgrad16') and rel_err(dd, ref) < 5e-5


def test_direct_wgrad_image_borders(dev):
    """Zero padding through out-of-range buffer offsets: a gradient that lives ONLY on the border pixels, an input that lives only on the
    border — every tap's shifted window crosses the image edge and the neighbouring image of the batch."""
    B, C_, H, W = 3, 64, 8, 16
    torch.manual_seed(3)
    x = torch.randn(B, C_, H, W, dtype=torch.float64)
    g = torch.zeros(B, C_, H, W, dtype=torch.float64)
    g[:, :, 0, :] = torch.randn(B, C_, W); g[:, :, -1, :] = torch.randn(B, C_, W)
    g[:, :, :, 0] = torch.randn(B, C_, H); g[:, :, :, -1] = torch.randn(B, C_, H)
    w = torch.zeros(C_, C_, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x, w, padding=1) * g).sum().backward()
    ref = w.grad.permute(0, 2, 3, 1).reshape(C_, 9, C_)
    for precision, tol in ((0, 3e-6), (1, 3e-5)):
        dw, sym = _run(nhwc(x.float()).to(dev), nhwc(g.float()).to(dev), C_, C_, precision)
        assert sym.startswith('wgrad_direct')
        assert rel_err(dw, ref) < tol
