"""GPU: the fp16-operand arithmetic modes of the implicit-GEMM conv kernels (wgs_conv_desc.precision 2 = 'f16', 3 = 'f16x2').

Two kinds of check per kernel family (patch form with 1 and 3 taps per barrier, register-staged 128/256-row tiles, split-K,
merged sub-pixel phases, LDS-DMA form, strided dgrad):
  * EXACTNESS of the implementation: against a float64 convolution of the SAME fp16-rounded operands
    (activation * style rounded to fp16 once, weights rounded to fp16 / fp16 hi+lo) — only fp32 accumulation order remains;
  * ACCURACY of the mode: against the float64 convolution of the unrounded operands (2^-11 per operand: ~3e-4 relative).
Plus the dynamic power-of-two operand scale (a_amax) that keeps tiny / huge gradient operands inside fp16's range."""
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


def r16(t):
    return t.float().half().double()


def r16x2(t):
    h = t.float().half()
    return h.double() + (t.float() - h.float()).half().double()


def wq(w, prec):
    return r16(w) if prec == 2 else r16x2(w)


@pytest.mark.parametrize('prec', [2, 3])
@pytest.mark.parametrize('B,Ci,Co,H,k,s,p', [(2, 32, 64, 9, 3, 1, 1), (3, 64, 128, 8, 3, 1, 1), (2, 128, 160, 6, 3, 1, 1),
                                             (2, 64, 128, 9, 3, 2, 1), (1, 512, 512, 4, 3, 1, 1), (2, 256, 24, 12, 1, 1, 0)])
def test_f16_small_shapes_fwd_dgrad(dev, prec, B, Ci, Co, H, k, s, p):
    torch.manual_seed(Ci + Co + prec)
    x = torch.randn(B, Ci, H, H, dtype=torch.float64)
    w = torch.randn(Co, Ci, k, k, dtype=torch.float64) / (Ci * k * k) ** 0.5
    sc = (torch.randn(B, Ci) + 1.0).double()
    xs = (x.float() * sc.float()[:, :, None, None]).double()          # the kernel's rounded fp32 product
    y_q = F.conv2d(r16(xs), wq(w, prec), stride=s, padding=p)
    y_x = F.conv2d(xs, w, stride=s, padding=p)
    xd = nhwc(x.float()).to(dev)
    wp = C.pack_weight(w.float()).to(dev)
    yd = C.conv2d(xd, wp, k, stride=s, pad=p, a_scale=sc.float().to(dev), precision=prec)
    e_impl, e_mode = rel_err(nchw(yd), y_q), rel_err(nchw(yd), y_x)
    # dgrad: gradient operand with a magnitude bound (dynamic scale) — tiny values on purpose
    g = torch.randn_like(y_x) * 3e-7
    amax = g.abs().max().float().reshape(1).to(dev)
    wt = C.repack_w_t(wp, Co, k * k, Ci)
    dx = C.conv2d_dgrad(nhwc(g.float()).to(dev), wt, (H, H), k, stride=s, pad=p, precision=prec, a_amax=amax)
    xr = torch.zeros(B, Ci, H, H, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xr, w, stride=s, padding=p) * g).sum().backward()
    e_dg = rel_err(nchw(dx), xr.grad)
    print('precision %d conv %s: implementation %.2e, mode fwd %.2e dgrad %.2e' % (prec, (B, Ci, Co, H, k, s, p), e_impl, e_mode, e_dg))
    assert e_impl < 2e-6
    assert e_mode < 2e-3 and e_dg < 2e-3
    # without a bound the gradient launch must not run in fp16 (it falls back to split-bf16): still accurate
    dx2 = C.conv2d_dgrad(nhwc(g.float()).to(dev), wt, (H, H), k, stride=s, pad=p, precision=prec)
    assert rel_err(nchw(dx2), xr.grad) < 1e-4


@pytest.mark.parametrize('prec', [2, 3])
@pytest.mark.parametrize('B,Ci,Co,H', [(8, 128, 128, 64), (2, 64, 128, 256), (32, 32, 256, 32), (32, 128, 512, 16),
                                       (16, 64, 128, 48), (32, 64, 256, 40)])
def test_f16_patch_form(dev, prec, B, Ci, Co, H, dev_flags):
    """Pre-split fp16 weights => patch kernel (TPS = 3 where the LDS budget allows, else 1): identical to the register-staged
    kernel of the same mode, identical between 3 and 1 taps per barrier, close to the exact kernel; one CPU sample."""
    torch.manual_seed(H + Co + prec)
    x = torch.randn(B, H, H, Ci, device=dev)
    w = torch.randn(Co, Ci, 3, 3) / (Ci * 9) ** 0.5
    wp = C.pack_weight(w).to(dev)
    ws = C.split_weight(wp, prec)
    assert torch.equal(ws[0].view(torch.float16).float(), wp.half().float())
    s = torch.randn(B, Ci, device=dev) + 1.0
    dm = torch.rand(B, Co, device=dev) + 0.5
    bias, noise, nw = torch.randn(Co, device=dev), torch.randn(H, H, device=dev), torch.tensor([0.37], device=dev)
    kw = dict(a_scale=s, col_scale=dm, bias=bias, noise=noise, noise_w=nw, act_slope=0.2, gain=2 ** 0.5)
    y_p = C.conv2d(x, wp, 3, pad=1, precision=prec, w_split=ws, **kw)
    y_r = C.conv2d(x, wp, 3, pad=1, precision=prec, **kw)
    assert torch.equal(y_p, y_r)
    dev_flags(WGS_PATCH_TPS1='1')
    y_1 = C.conv2d(x, wp, 3, pad=1, precision=prec, w_split=ws, **kw)
    assert torch.equal(y_p, y_1)
    e = rel_err(y_p, C.conv2d(x, wp, 3, pad=1, precision=0, **kw))
    xs = (x[:1] * s[:1, None, None, :]).cpu().permute(0, 3, 1, 2)
    ref = F.conv2d(r16(xs), wq(w.double(), prec), padding=1) * dm[:1].cpu().double()[:, :, None, None] \
        + (nw.cpu() * noise.cpu()).double()[None, None] + bias.cpu().double()[None, :, None, None]
    e_impl = rel_err(nchw(y_p[:1]), F.leaky_relu(ref, 0.2) * 2 ** 0.5)
    print('precision %d patch %s: vs exact kernel %.2e, vs fp64 of the rounded operands %.2e' % (prec, (B, Ci, Co, H), e, e_impl))
    assert e < 2e-3 and e_impl < 3e-6
    if Ci % 128 == 0:
        wt = C.repack_w_t(wp, Co, 9, Ci)
        g = torch.randn(B, H, H, Co, device=dev) * 1e-6
        am = g.abs().max().reshape(1)
        d_p = C.conv2d_dgrad(g, wt, (H, H), 3, pad=1, precision=prec, w_split=C.split_weight(wt, prec), a_amax=am)
        assert torch.equal(d_p, C.conv2d_dgrad(g, wt, (H, H), 3, pad=1, precision=prec, a_amax=am))
        assert rel_err(d_p, C.conv2d_dgrad(g, wt, (H, H), 3, pad=1, precision=0)) < 2e-3


@pytest.mark.parametrize('prec', [2, 3])
@pytest.mark.parametrize('B,Ci,Co,H', [(4, 64, 256, 128), (8, 32, 128, 128), (8, 32, 512, 64)])
def test_f16_lds_dma_form_and_256_row_tiles(dev, prec, B, Ci, Co, H, dev_flags):
    dev_flags(WGS_DMA_ALWAYS='1', WGS_NO_PATCH='1')
    torch.manual_seed(Co + prec)
    x = torch.randn(B, H, H, Ci, device=dev)
    wp = C.pack_weight(torch.randn(Co, Ci, 3, 3) / (Ci * 9) ** 0.5).to(dev)
    ws = C.split_weight(wp, prec)
    s = torch.randn(B, Ci, device=dev) + 1.0
    dm = torch.rand(B, Co, device=dev) + 0.5
    kw = dict(a_scale=s, col_scale=dm, act_slope=0.2, gain=2 ** 0.5)
    y_dma = C.conv2d(x, wp, 3, pad=1, precision=prec, w_split=ws, **kw)      # pre-pass + DMA kernel
    y_reg = C.conv2d(x, wp, 3, pad=1, precision=prec, **kw)                  # register-staged 8-wave tiles
    assert torch.equal(y_dma, y_reg)
    assert rel_err(y_dma, C.conv2d(x, wp, 3, pad=1, precision=0, **kw)) < 2e-3
    assert torch.equal(C.conv2d(x, wp, 3, pad=1, precision=prec, w_split=ws), C.conv2d(x, wp, 3, pad=1, precision=prec))


@pytest.mark.parametrize('prec', [2, 3])
@pytest.mark.parametrize('B,Ci,Co,H', [(2, 64, 128, 6), (1, 512, 512, 4), (8, 64, 128, 64), (4, 128, 512, 32)])
def test_f16_transposed_conv_and_its_dgrad(dev, prec, B, Ci, Co, H):
    """Sub-pixel phase launches (split-K for the small maps, the merged 4-phase launch for the large ones, LDS-DMA for
    Cout >= 512) and the stride-2 dgrad of the transposed conv, with a scaled gradient operand."""
    torch.manual_seed(H + Ci + prec)
    x = torch.randn(B, Ci, H, H, dtype=torch.float64)
    w = torch.randn(Ci, Co, 3, 3, dtype=torch.float64) / (Ci * 9) ** 0.5
    s = (torch.randn(B, Ci) + 1.0).double()
    xs = (x.float() * s.float()[:, :, None, None]).double()
    y_q = F.conv_transpose2d(r16(xs), wq(w, prec), stride=2)
    wp = C.pack_weight(w.permute(1, 0, 2, 3).float()).to(dev)
    xd = nhwc(x.float()).to(dev)
    ys = C.conv_transpose2d_s2(xd, wp, a_scale=s.float().to(dev), precision=prec)
    ys2 = C.conv_transpose2d_s2(xd, wp, a_scale=s.float().to(dev), precision=prec, w_split=C.split_weight(wp, prec))
    assert rel_err(nchw(ys), y_q) < 3e-6
    assert torch.equal(ys, ys2)
    assert rel_err(nchw(ys), F.conv_transpose2d(xs, w, stride=2)) < 2e-3
    g = torch.randn(B, Co, 2 * H + 1, 2 * H + 1, dtype=torch.float64) * 2e-5
    am = (g.abs().max() / 4).float().reshape(1).to(dev)          # bound given as max/4 with a_bound = 4 (the blur's gain)
    wt = C.repack_w_t(wp, Co, 9, Ci)
    dx = C.conv_transpose2d_s2_dgrad(nhwc(g.float()).to(dev), wt, precision=prec, a_amax=am, a_bound=4.0,
                                     w_split=C.split_weight(wt, prec))
    xr = x.clone().requires_grad_(True)
    (F.conv_transpose2d(xr, w, stride=2) * g).sum().backward()
    e = rel_err(nchw(dx), xr.grad)
    print('precision %d transposed conv %s: dgrad %.2e' % (prec, (B, Ci, Co, H), e))
    assert e < 2e-3


def test_f16_dynamic_operand_scale_range(dev):
    """Gradient operands from 1e-30 to 1e+30: with the device-resident bound the fp16 launch keeps its relative accuracy;
    magnitudes outside fp16's range would otherwise flush to zero / overflow."""
    torch.manual_seed(3)
    B, Ci, Co, H = 4, 128, 128, 32
    wp = C.pack_weight(torch.randn(Co, Ci, 3, 3) / (Ci * 9) ** 0.5).to(dev)
    ws = C.split_weight(wp, 2)
    g0 = torch.randn(B, H, H, Ci, device=dev)
    ref = C.conv2d(g0, wp, 3, pad=1, precision=0)
    for mag in (1e-30, 1e-12, 1e-6, 1.0, 1e4, 1e12, 1e30):
        g = g0 * mag
        am = g.abs().max().reshape(1)
        y = C.conv2d(g, wp, 3, pad=1, precision=2, w_split=ws, a_amax=am)
        e = rel_err(y / mag, ref)
        assert e < 1e-3, (mag, e)
        y2 = C.conv2d(g, wp, 3, pad=1, precision=2, a_amax=am * 3.0)      # any over-estimate works (register-staged kernel)
        assert rel_err(y2 / mag, ref) < 1e-3
    tiny = C.conv2d(g0 * 1e-9, wp, 3, pad=1, precision=2, w_split=ws)     # no bound: below fp16's subnormals -> all zero
    assert float(tiny.abs().max()) == 0.0


def test_f16_patch_kernel_is_stable_across_repeated_launches(dev):
    """Regression (round 2): a counted s_waitcnt that left the patch loads in flight across the weight-stage barrier relied on
    LDS-DMA and VGPR loads retiring in issue order; they do not, and the 128->128 @256x256 gradient conv (shortest K loop, two
    workgroups per CU) read stale weight stages from its second launch on.  Every launch must reproduce the first one."""
    torch.manual_seed(0)
    for B, ci, co, h in [(2, 128, 128, 256), (8, 128, 128, 256), (2, 256, 256, 128)]:
        w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
        wt = C.repack_w_t(w, co, 9, ci)
        ws = C.split_weight(wt, 2)
        g = torch.randn(B, h, h, co, device=dev) * 1e-5
        am = g.abs().max().reshape(1)
        ref = C.conv2d_dgrad(g, wt, (h, h), 3, pad=1, precision=1)
        first = None
        for rep in range(6):
            d = C.conv2d_dgrad(g, wt, (h, h), 3, pad=1, precision=2, w_split=ws, a_amax=am)
            assert rel_err(d, ref) < 1e-3, (B, ci, h, rep, rel_err(d, ref))
            if first is None:
                first = d
            assert torch.equal(d, first), (B, ci, h, rep)
