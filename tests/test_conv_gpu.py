"""GPU: implicit-GEMM MFMA convolution kernels vs torch-CPU fp64 convolution (the oracle's op)."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_err
from warpedganspace_amd import conv as C

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


CASES = [  # B, Ci, Co, H, W, k, stride, pad
    (2, 32, 64, 9, 7, 3, 1, 1),
    (3, 64, 128, 8, 8, 3, 1, 1),
    (2, 128, 160, 6, 6, 3, 1, 1),      # Co not a multiple of the tile
    (2, 64, 128, 9, 9, 3, 2, 1),
    (2, 64, 128, 8, 8, 1, 2, 0),
    (2, 8, 64, 20, 20, 7, 2, 3),       # ResNet conv1 shape (6 channels padded to 8)
    (1, 512, 512, 4, 4, 3, 1, 1),
    (2, 40, 24, 5, 5, 5, 1, 0),        # LeNet-like 5x5, Ci % 8 == 0
    (5, 32, 32, 3, 3, 3, 1, 1),
]


@pytest.mark.parametrize('B,Ci,Co,H,W,k,s,p', CASES)
def test_conv_fwd_dgrad_wgrad(dev, B, Ci, Co, H, W, k, s, p):
    torch.manual_seed(B * 1000 + Ci + Co)
    x = torch.randn(B, Ci, H, W, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Co, Ci, k, k, dtype=torch.float64) / (Ci * k * k) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, stride=s, padding=p)
    g = torch.randn_like(y)
    (y * g).sum().backward()

    xd = nhwc(x.detach().float()).to(dev)
    wp = C.pack_weight(w.detach().float()).to(dev)
    yd = C.conv2d(xd, wp, k, stride=s, pad=p)
    assert rel_err(nchw(yd), y.detach()) < 1e-5
    gd = nhwc(g.float()).to(dev)
    wt = C.repack_w_t(wp, Co, k * k, Ci)
    assert torch.equal(wt.cpu(), wp.cpu().permute(1, 2, 0).contiguous())
    dx = C.conv2d_dgrad(gd, wt, (H, W), k, stride=s, pad=p)
    assert rel_err(nchw(dx), x.grad) < 1e-5
    if Ci % 4 == 0 and Co % 4 == 0:
        dw_ref = w.grad.permute(0, 2, 3, 1).reshape(Co, k * k, Ci)
        dw = torch.zeros_like(wp)
        C.conv2d_wgrad(xd, gd, dw, k, stride=s, pad=p)
        assert rel_err(dw, dw_ref) < 2e-5
        dw1 = torch.zeros_like(wp)
        C.conv2d_wgrad(xd, gd, dw1, k, stride=s, pad=p, ksplit=1)
        assert rel_err(dw1, dw_ref) < 2e-5


def test_conv_epilogue_and_prologue(dev):
    """StyleGAN2 fusion: a_scale (style), col_scale (demod), noise, bias, leaky-relu*sqrt(2)."""
    torch.manual_seed(5)
    B, Ci, Co, H = 3, 64, 96, 8
    x = torch.randn(B, Ci, H, H)
    w = torch.randn(Co, Ci, 3, 3) / (Ci * 9) ** 0.5
    s = torch.randn(B, Ci) + 1.0
    dm = torch.rand(B, Co) + 0.5
    bias = torch.randn(Co)
    noise = torch.randn(H, H)
    nw = torch.tensor([0.37])
    ref = (F.conv2d(x * s[:, :, None, None], w, padding=1) * dm[:, :, None, None] + nw * noise[None, None]
           + bias[None, :, None, None])
    ref = F.leaky_relu(ref, 0.2) * 2 ** 0.5
    y = C.conv2d(nhwc(x).to(dev), C.pack_weight(w).to(dev), 3, pad=1, a_scale=s.to(dev), col_scale=dm.to(dev),
                 bias=bias.to(dev), noise=noise.to(dev), noise_w=nw.to(dev), act_slope=0.2, gain=2 ** 0.5)
    assert rel_err(nchw(y), ref) < 1e-5


@pytest.mark.parametrize('B,Ci,Co,H', [(2, 64, 32, 5), (1, 32, 64, 4), (3, 128, 128, 8)])
def test_conv_transpose_s2(dev, B, Ci, Co, H):
    torch.manual_seed(H)
    x = torch.randn(B, Ci, H, H, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Ci, Co, 3, 3, dtype=torch.float64) / (Ci * 9) ** 0.5   # conv_transpose2d layout [in,out,k,k]
    y = F.conv_transpose2d(x, w, stride=2, padding=0)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    wp = C.pack_weight(w.permute(1, 0, 2, 3).float()).to(dev)     # [Co, 9, Ci]
    yd = C.conv_transpose2d_s2(nhwc(x.detach().float()).to(dev), wp)
    assert tuple(yd.shape) == (B, 2 * H + 1, 2 * H + 1, Co)
    assert rel_err(nchw(yd), y.detach()) < 1e-5
    wt = C.repack_w_t(wp, Co, 9, Ci)
    dx = C.conv_transpose2d_s2_dgrad(nhwc(g.float()).to(dev), wt)
    assert rel_err(nchw(dx), x.grad) < 1e-5


def test_conv_large_matches_blockwise(dev):
    """A BASELINE-size layer (128->128 @ 64x64, B=4): compare with the CPU conv on one sample, and
    linearity conv(a*x1 + x2) = a*conv(x1) + conv(x2) on the full tensor."""
    torch.manual_seed(1)
    B, Cc, H = 4, 128, 64
    w = torch.randn(Cc, Cc, 3, 3) / (Cc * 9) ** 0.5
    x1, x2 = torch.randn(B, H, H, Cc, device=dev), torch.randn(B, H, H, Cc, device=dev)
    wp = C.pack_weight(w).to(dev)
    y1, y2 = C.conv2d(x1, wp, 3, pad=1), C.conv2d(x2, wp, 3, pad=1)
    y3 = C.conv2d(0.5 * x1 + x2, wp, 3, pad=1)
    assert rel_err(y3, 0.5 * y1 + y2) < 1e-5
    ref = F.conv2d(x1[:1].cpu().permute(0, 3, 1, 2), w, padding=1)
    assert rel_err(nchw(y1[:1]), ref) < 1e-5


@pytest.mark.parametrize('B,Ci,Co,H,k,s,p', [(2, 32, 64, 9, 3, 1, 1), (3, 64, 128, 8, 3, 1, 1), (2, 128, 160, 6, 3, 1, 1),
                                             (2, 64, 128, 9, 3, 2, 1), (1, 512, 512, 4, 3, 1, 1), (2, 256, 24, 12, 1, 1, 0)])
def test_conv_split_bf16_precision(dev, B, Ci, Co, H, k, s, p):
    """precision=1 (3 x bf16 MFMA on hi/lo splits): fwd / dgrad within ~2e-5 of float64 — fp32-class, inside the 1e-3 gate."""
    torch.manual_seed(Ci + Co)
    x = torch.randn(B, Ci, H, H, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Co, Ci, k, k, dtype=torch.float64) / (Ci * k * k) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, stride=s, padding=p)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    xd = nhwc(x.detach().float()).to(dev)
    wp = C.pack_weight(w.detach().float()).to(dev)
    yd = C.conv2d(xd, wp, k, stride=s, pad=p, precision=1)
    e1 = rel_err(nchw(yd), y.detach())
    dx = C.conv2d_dgrad(nhwc(g.float()).to(dev), C.repack_w_t(wp, Co, k * k, Ci), (H, H), k, stride=s, pad=p, precision=1)
    e2 = rel_err(nchw(dx), x.grad)
    print('split-bf16 conv %s: fwd %.2e dgrad %.2e' % ((B, Ci, Co, H, k, s, p), e1, e2))
    assert e1 < 1e-4 and e2 < 1e-4
    yx = C.conv2d(xd, wp, k, stride=s, pad=p, precision=0)
    assert rel_err(yd, yx) < 1e-4


def test_conv_split_bf16_fused_epilogue(dev):
    torch.manual_seed(6)
    B, Ci, Co, H = 3, 64, 96, 8
    x = torch.randn(B, Ci, H, H)
    w = torch.randn(Co, Ci, 3, 3) / (Ci * 9) ** 0.5
    s = torch.randn(B, Ci) + 1.0
    dm = torch.rand(B, Co) + 0.5
    bias, noise, nw = torch.randn(Co), torch.randn(H, H), torch.tensor([0.37])
    ref = (F.conv2d(x * s[:, :, None, None], w, padding=1) * dm[:, :, None, None] + nw * noise[None, None] + bias[None, :, None, None])
    ref = F.leaky_relu(ref, 0.2) * 2 ** 0.5
    y = C.conv2d(nhwc(x).to(dev), C.pack_weight(w).to(dev), 3, pad=1, a_scale=s.to(dev), col_scale=dm.to(dev), bias=bias.to(dev),
                 noise=noise.to(dev), noise_w=nw.to(dev), act_slope=0.2, gain=2 ** 0.5, precision=1)
    assert rel_err(nchw(y), ref) < 1e-4


@pytest.mark.parametrize('B,Ci,Co,H', [(2, 64, 128, 6), (4, 128, 256, 8), (1, 512, 512, 4)])
def test_conv_transpose_s2_split_bf16(dev, B, Ci, Co, H):
    """precision=1 on the sub-pixel phase launches of the 4x4..16x16 generator layers: few tiles, so the K contraction is
    split over workgroups (wgs_conv_desc.ws) and finished by the reduction/epilogue kernel, with strided phase outputs."""
    torch.manual_seed(H + Ci)
    x = torch.randn(B, Ci, H, H, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Ci, Co, 3, 3, dtype=torch.float64) / (Ci * 9) ** 0.5
    y = F.conv_transpose2d(x, w, stride=2, padding=0)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    wp = C.pack_weight(w.permute(1, 0, 2, 3).float()).to(dev)
    s = (torch.randn(B, Ci) + 1.0)
    yd = C.conv_transpose2d_s2(nhwc(x.detach().float()).to(dev), wp, precision=1)
    assert rel_err(nchw(yd), y.detach()) < 1e-4
    ys = C.conv_transpose2d_s2(nhwc(x.detach().float()).to(dev), wp, a_scale=s.to(dev), precision=1)
    ref_s = F.conv_transpose2d(x.detach() * s.double()[:, :, None, None], w, stride=2, padding=0)
    assert rel_err(nchw(ys), ref_s) < 1e-4
    dx = C.conv_transpose2d_s2_dgrad(nhwc(g.float()).to(dev), C.repack_w_t(wp, Co, 9, Ci), precision=1)
    assert rel_err(nchw(dx), x.grad) < 1e-4


@pytest.mark.parametrize('B,Ci,Co,H', [(4, 64, 256, 128), (8, 32, 128, 128), (2, 64, 512, 256)])
def test_conv_split_bf16_256_row_tiles(dev, B, Ci, Co, H):
    """Shapes that take the 8-wave 256x256 / 256x128 tiles (enough 256-row tiles to fill the chip), with the whole
    StyleGAN2 epilogue; reference = the exact-fp32 kernel (itself checked against PyTorch above) plus one CPU sample."""
    torch.manual_seed(Co)
    x = torch.randn(B, H, H, Ci, device=dev)
    w = torch.randn(Co, Ci, 3, 3) / (Ci * 9) ** 0.5
    wp = C.pack_weight(w).to(dev)
    s = torch.randn(B, Ci, device=dev) + 1.0
    dm = torch.rand(B, Co, device=dev) + 0.5
    bias, noise, nw = torch.randn(Co, device=dev), torch.randn(H, H, device=dev), torch.tensor([0.37], device=dev)
    kw = dict(a_scale=s, col_scale=dm, bias=bias, noise=noise, noise_w=nw, act_slope=0.2, gain=2 ** 0.5)
    y1 = C.conv2d(x, wp, 3, pad=1, precision=1, **kw)
    y0 = C.conv2d(x, wp, 3, pad=1, precision=0, **kw)
    assert rel_err(y1, y0) < 2e-5
    xb = x[B - 1:].cpu().permute(0, 3, 1, 2).double() * s[B - 1:].cpu().double()[:, :, None, None]
    ref = F.conv2d(xb, w.double(), padding=1) * dm[B - 1:].cpu().double()[:, :, None, None] + (nw.cpu() * noise.cpu()).double()[None, None] \
        + bias.cpu().double()[None, :, None, None]
    ref = F.leaky_relu(ref, 0.2) * 2 ** 0.5
    assert rel_err(nchw(y1[B - 1:]), ref) < 2e-5
    y2 = C.conv2d(x, wp, 3, pad=1, precision=1)            # no style: the ASCALE=0 instantiation
    assert rel_err(y2, C.conv2d(x, wp, 3, pad=1, precision=0)) < 2e-5


@pytest.mark.parametrize('B,Ci,Co,H', [(4, 64, 256, 128), (8, 32, 128, 128), (8, 32, 512, 64)])
def test_conv_split_bf16_lds_dma_path(dev, B, Ci, Co, H, dev_flags):
    """Pre-split weights (wgs_split_bf16) + workspace => the LDS-DMA kernel: same split values, same product order as the
    register-staged kernel, so the results must be identical; also against the exact-fp32 kernel.  (The library takes the
    DMA form on its own only for Cout >= 512, where it pays; WGS_DMA_ALWAYS forces it for the other 8-wave shapes.)"""
    dev_flags(WGS_DMA_ALWAYS='1')
    torch.manual_seed(Co + 1)
    x = torch.randn(B, H, H, Ci, device=dev)
    wp = C.pack_weight(torch.randn(Co, Ci, 3, 3) / (Ci * 9) ** 0.5).to(dev)
    ws = C.split_weight(wp)
    hi = ws[0].view(torch.bfloat16).float()
    assert torch.equal(hi, wp.to(torch.bfloat16).float())                       # round-to-nearest-even split
    assert torch.equal(ws[1].view(torch.bfloat16).float(), (wp - hi).to(torch.bfloat16).float())
    s = torch.randn(B, Ci, device=dev) + 1.0
    dm = torch.rand(B, Co, device=dev) + 0.5
    bias, noise, nw = torch.randn(Co, device=dev), torch.randn(H, H, device=dev), torch.tensor([0.37], device=dev)
    kw = dict(a_scale=s, col_scale=dm, bias=bias, noise=noise, noise_w=nw, act_slope=0.2, gain=2 ** 0.5)
    y_dma = C.conv2d(x, wp, 3, pad=1, precision=1, w_split=ws, **kw)
    y_reg = C.conv2d(x, wp, 3, pad=1, precision=1, **kw)
    assert torch.equal(y_dma, y_reg)
    assert rel_err(y_dma, C.conv2d(x, wp, 3, pad=1, precision=0, **kw)) < 2e-5
    y2 = C.conv2d(x, wp, 3, pad=1, precision=1, w_split=ws)                     # no style
    assert torch.equal(y2, C.conv2d(x, wp, 3, pad=1, precision=1))
    # dgrad form (weights packed [T, Ci, Co]) with a row scale
    wt = C.repack_w_t(wp, Co, 9, Ci)
    g = torch.randn(B, H, H, Co, device=dev)
    d1 = C.conv2d_dgrad(g, wt, (H, H), 3, pad=1, a_scale=dm, precision=1, w_split=C.split_weight(wt)) if Ci % 128 == 0 else None
    if d1 is not None:
        assert torch.equal(d1, C.conv2d_dgrad(g, wt, (H, H), 3, pad=1, a_scale=dm, precision=1))


def test_conv_transpose_s2_merged_phases_lds_dma(dev, dev_flags):
    """Up-conv large enough for the merged 4-phase launch, with and without pre-split weights, vs the exact-fp32 phases."""
    dev_flags(WGS_DMA_ALWAYS='1')
    torch.manual_seed(11)
    B, Ci, Co, H = 8, 64, 128, 64
    x = torch.randn(B, H, H, Ci, device=dev)
    wp = C.pack_weight(torch.randn(Co, Ci, 3, 3) / (Ci * 9) ** 0.5).to(dev)
    s = torch.randn(B, Ci, device=dev) + 1.0
    dm = torch.rand(B, Co, device=dev) + 0.5
    y0 = C.conv_transpose2d_s2(x, wp, a_scale=s, col_scale=dm, precision=0)
    y1 = C.conv_transpose2d_s2(x, wp, a_scale=s, col_scale=dm, precision=1)
    y2 = C.conv_transpose2d_s2(x, wp, a_scale=s, col_scale=dm, precision=1, w_split=C.split_weight(wp))
    assert rel_err(y1, y0) < 2e-5
    assert torch.equal(y1, y2)
    ref = F.conv_transpose2d((x[:1].cpu().permute(0, 3, 1, 2) * s[:1].cpu()[:, :, None, None]).double(),
                             wp.cpu().reshape(Co, 3, 3, Ci).permute(3, 0, 1, 2).double(), stride=2) * dm[:1].cpu().double()[:, :, None, None]
    assert rel_err(nchw(y2[:1]), ref) < 2e-5


@pytest.mark.parametrize('B,Ci,Co,H', [(8, 128, 128, 64), (2, 64, 128, 256), (32, 32, 256, 32), (32, 128, 512, 16),
                                       (16, 64, 128, 48), (32, 64, 256, 40)])    # last two: grids that are not powers of two
def test_conv_split_bf16_patch_form(dev, B, Ci, Co, H):
    """Stride-1 3x3 convs with pre-split weights take the patch kernel (halo patch staged once per channel chunk, taps read
    shifted rows): forward and dgrad, all tile geometries (2x128, 4x64, 8x32 pixels), identical to the register-staged
    kernel and within 2e-5 of the exact one; plus one CPU sample."""
    torch.manual_seed(H + Co)
    x = torch.randn(B, H, H, Ci, device=dev)
    w = torch.randn(Co, Ci, 3, 3) / (Ci * 9) ** 0.5
    wp = C.pack_weight(w).to(dev)
    ws = C.split_weight(wp)
    s = torch.randn(B, Ci, device=dev) + 1.0
    dm = torch.rand(B, Co, device=dev) + 0.5
    bias, noise, nw = torch.randn(Co, device=dev), torch.randn(H, H, device=dev), torch.tensor([0.37], device=dev)
    kw = dict(a_scale=s, col_scale=dm, bias=bias, noise=noise, noise_w=nw, act_slope=0.2, gain=2 ** 0.5)
    y_p = C.conv2d(x, wp, 3, pad=1, precision=1, w_split=ws, **kw)
    y_r = C.conv2d(x, wp, 3, pad=1, precision=1, **kw)
    assert torch.equal(y_p, y_r)
    assert rel_err(y_p, C.conv2d(x, wp, 3, pad=1, precision=0, **kw)) < 2e-5
    xb = x[:1].cpu().permute(0, 3, 1, 2).double() * s[:1].cpu().double()[:, :, None, None]
    ref = F.conv2d(xb, w.double(), padding=1) * dm[:1].cpu().double()[:, :, None, None] + (nw.cpu() * noise.cpu()).double()[None, None] \
        + bias.cpu().double()[None, :, None, None]
    assert rel_err(nchw(y_p[:1]), F.leaky_relu(ref, 0.2) * 2 ** 0.5) < 2e-5
    if Ci % 128 == 0:
        wt = C.repack_w_t(wp, Co, 9, Ci)
        g = torch.randn(B, H, H, Co, device=dev)
        d_p = C.conv2d_dgrad(g, wt, (H, H), 3, pad=1, a_scale=dm, precision=1, w_split=C.split_weight(wt))
        assert torch.equal(d_p, C.conv2d_dgrad(g, wt, (H, H), 3, pad=1, a_scale=dm, precision=1))
        assert rel_err(d_p, C.conv2d_dgrad(g, wt, (H, H), 3, pad=1, a_scale=dm, precision=0)) < 2e-5


@pytest.mark.parametrize('B,Ci,Co,H,k,s,p', [(4, 64, 64, 16, 3, 1, 1), (3, 64, 128, 17, 3, 2, 1), (2, 128, 128, 9, 3, 1, 1),
                                             (2, 256, 512, 8, 1, 2, 0), (5, 128, 64, 7, 3, 1, 1), (32, 64, 64, 64, 3, 1, 1),
                                             (2, 512, 512, 8, 3, 1, 1), (3, 64, 128, 16, 3, 1, 1), (2, 128, 64, 24, 3, 1, 1)])
def test_conv_wgrad_split_bf16(dev, B, Ci, Co, H, k, s, p):
    """wgs_wgrad_desc.precision = 1: operands transposed in registers into the [channel][pixel] LDS image and split into
    bf16 hi / lo, 3 MFMAs per product — within ~2e-5 of float64 (the exact kernel: ~1e-6), any K split.  Covers the per-tap
    kernel and the kernel-row form (stride-1 3x3, width % 8 == 0, 64 or >= 512 channels: three taps per workgroup)."""
    torch.manual_seed(Ci + Co + H)
    x = torch.randn(B, Ci, H, H, dtype=torch.float64)
    w = (torch.randn(Co, Ci, k, k, dtype=torch.float64) / (Ci * k * k) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, stride=s, padding=p)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    dw_ref = w.grad.permute(0, 2, 3, 1).reshape(Co, k * k, Ci)
    xd, gd = nhwc(x.float()).to(dev), nhwc(g.float()).to(dev)
    for ksplit in (0, 1, 3):
        dw = torch.zeros(Co, k * k, Ci, device=dev)
        C.conv2d_wgrad(xd, gd, dw, k, stride=s, pad=p, ksplit=ksplit, precision=1)
        e = rel_err(dw, dw_ref)
        assert e < 5e-5, (ksplit, e)
    dw0 = torch.zeros(Co, k * k, Ci, device=dev)
    C.conv2d_wgrad(xd, gd, dw0, k, stride=s, pad=p, precision=0)
    assert rel_err(dw, dw0) < 5e-5
