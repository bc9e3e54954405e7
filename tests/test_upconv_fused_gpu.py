"""GPU: the fused StyleGAN2 up-sampling layer (wgs_sg2_upconv_blur_act, conv_upfused.hip) — modulated conv_transpose2d
stride 2 + Blur(pad (1,1)) + noise + bias + leaky-relu*sqrt(2) in one launch (models/StyleGAN2/model.py:201-212,231-241,264).

  * EXACTNESS: against float64 conv_transpose2d + upfirdn blur of the SAME fp16-rounded operands (only the fp32
    accumulation order and the fp32 blur remain): ~1e-6;
  * ACCURACY of the mode against the unrounded float64 layer: the fp16 operand rounding, ~3e-4;
  * agreement with the unfused launches (phase GEMMs + wgs_sg2_blur_noise_bias_act): bit-level for fp16; for fp16 x2 the
    fused kernel splits the activation operand where the GEMM kernels split the weights (same error class);
  * sizes: tile interior / border / image smaller than a tile / image not a multiple of the 14-cell tile / both weight planes."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_err
from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C

pytestmark = pytest.mark.gpu

SQRT2 = 2.0 ** 0.5


def r16(t):
    return t.float().half().double()


def r16x2(t):
    h = t.float().half()
    return h.double() + (t.float() - h.float()).half().double()


def blur_kernel():
    k = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=torch.float64)
    k2 = k[:, None] * k[None, :]
    return k2 / k2.sum() * 4.0                      # Blur(kernel, upsample_factor=2): kernel * factor^2 (model.py:76-77)


def layer_f64(xs, w, demod, kern, noise, nw, bias):
    """xs [B,Ci,H,H] (already style-modulated), w [Co,Ci,3,3]: the layer in float64."""
    t = F.conv_transpose2d(xs, w.transpose(0, 1), stride=2, padding=0)            # [B,Co,2H+1,2H+1]
    t = t * demod[:, :, None, None]
    Co = t.shape[1]
    tp = F.pad(t, (1, 1, 1, 1))
    kf = torch.flip(kern, [0, 1])[None, None].repeat(Co, 1, 1, 1)
    y = F.conv2d(tp, kf, groups=Co)                                               # upfirdn2d: correlate with the flipped kernel
    y = y + nw * noise[None, None] + bias[None, :, None, None]
    return F.leaky_relu(y, 0.2) * SQRT2


@pytest.mark.parametrize('prec', [2, 3])
@pytest.mark.parametrize('B,Ci,Co,H', [(2, 32, 64, 16), (1, 64, 128, 14), (2, 64, 64, 20), (1, 96, 64, 33), (3, 32, 128, 5), (2, 64, 32, 24), (1, 32, 32, 40)])
def test_upconv_fused_vs_float64(dev, prec, B, Ci, Co, H):
    torch.manual_seed(Ci * 7 + Co + H + prec)
    x = torch.randn(B, Ci, H, H, dtype=torch.float64)
    w = torch.randn(Co, Ci, 3, 3, dtype=torch.float64) / (Ci * 9) ** 0.5
    sc = (torch.randn(B, Ci) + 1.0).double()
    demod = (torch.rand(B, Co) + 0.5).double()
    noise = torch.randn(2 * H, 2 * H, dtype=torch.float64)
    nw = 0.3
    bias = torch.randn(Co, dtype=torch.float64) * 0.2
    kern = blur_kernel()
    xs = (x.float() * sc.float()[:, :, None, None]).double()              # the kernel's rounded fp32 product
    # precision 3 in THIS kernel: the activation operand carries the hi + lo split, the weights one fp16 plane (Scheme<3>)
    xq = r16(xs) if prec == 2 else r16x2(xs)
    y_q = layer_f64(xq, r16(w), demod.float().double(), kern.float().double(), noise.float().double(), nw, bias.float().double())
    y_x = layer_f64(xs, w, demod, kern, noise, nw, bias)

    xd = x.float().permute(0, 2, 3, 1).contiguous().to(dev)
    wp = C.pack_weight(w.float()).to(dev)
    ws = C.split_weight(wp, prec)
    S = sc.float().to(dev).contiguous()
    amax = torch.zeros(1, device=dev)
    y = C.upconv_blur_act(xd, ws, kern.float().to(dev), S, Ci, demod.float().to(dev), noise.float().reshape(-1).to(dev),
                          torch.tensor([nw], device=dev), bias.float().to(dev), prec, y_amax=amax)
    yn = y.permute(0, 3, 1, 2)
    e_impl, e_mode = rel_err(yn, y_q), rel_err(yn, y_x)
    assert e_impl < 3e-6, e_impl
    assert e_mode < 1.2e-3, e_mode
    assert abs(amax.item() - y.abs().max().item()) <= 1e-6 * amax.item()

    # the unfused launches of the same arithmetic
    dm_d, k_d, nz_d = demod.float().to(dev), kern.float().to(dev), noise.float().reshape(-1).to(dev)
    nw_d, b_d = torch.tensor([nw], device=dev), bias.float().to(dev)
    t = C.conv_transpose2d_s2(xd, wp, a_scale=S, a_ld=Ci, col_scale=dm_d, w_split=ws, precision=prec)
    y2 = torch.empty_like(y)
    L.check(L.lib().wgs_sg2_blur_noise_bias_act(L.ptr(t), L.ptr(k_d), L.ptr(nz_d), L.ptr(nw_d), L.ptr(b_d), L.ptr(y2), None,
                                                B, 2 * H, 2 * H, Co, L.stream()), 'blur_nba')
    # (fp16 x2: the unfused GEMMs split the WEIGHT operand instead — same error class, different rounding)
    assert rel_err(y, y2.cpu()) < (3e-6 if prec == 2 else 1.2e-3)


def test_upconv_fused_general_kernel(dev):
    """A 4x4 FIR that is NOT an outer product takes the kernel's 16-tap path (the separable one is a uniform branch)."""
    torch.manual_seed(11)
    B, Ci, Co, H = 2, 64, 64, 19
    kern = torch.rand(4, 4, dtype=torch.float64) + 0.1
    x = torch.randn(B, Ci, H, H, dtype=torch.float64)
    w = torch.randn(Co, Ci, 3, 3, dtype=torch.float64) / (Ci * 9) ** 0.5
    sc = (torch.randn(B, Ci) + 1.0).double()
    demod = (torch.rand(B, Co) + 0.5).double()
    noise = torch.randn(2 * H, 2 * H, dtype=torch.float64)
    bias = torch.randn(Co, dtype=torch.float64) * 0.2
    xs = (x.float() * sc.float()[:, :, None, None]).double()
    y_q = layer_f64(r16(xs), r16(w), demod.float().double(), kern.float().double(), noise.float().double(), 0.3, bias.float().double())
    wp = C.pack_weight(w.float()).to(dev)
    y = C.upconv_blur_act(x.float().permute(0, 2, 3, 1).contiguous().to(dev), C.split_weight(wp, 2), kern.float().to(dev),
                          sc.float().to(dev), Ci, demod.float().to(dev), noise.float().reshape(-1).to(dev),
                          torch.tensor([0.3], device=dev), bias.float().to(dev), 2)
    assert rel_err(y.permute(0, 3, 1, 2), y_q) < 3e-6


def test_upconv_fused_operand_scale(dev):
    """Activations far outside fp16's range: the magnitude chain (a_amax x a_amax2) keeps all 11 bits."""
    torch.manual_seed(5)
    B, Ci, Co, H = 1, 64, 64, 16
    kern = blur_kernel()
    w = torch.randn(Co, Ci, 3, 3, dtype=torch.float64) / (Ci * 9) ** 0.5
    demod = torch.ones(B, Co, dtype=torch.float64)
    noise = torch.zeros(2 * H, 2 * H, dtype=torch.float64)
    bias = torch.zeros(Co, dtype=torch.float64)
    for mag in (3e-7, 1.0, 4e6):
        x = torch.randn(B, Ci, H, H, dtype=torch.float64) * mag
        sc = (torch.rand(B, Ci) + 0.5).double()
        xs = (x.float() * sc.float()[:, :, None, None]).double()
        y_x = layer_f64(xs, w, demod, kern, noise, 0.0, bias)
        xd = x.float().permute(0, 2, 3, 1).contiguous().to(dev)
        wp = C.pack_weight(w.float()).to(dev)
        y = C.upconv_blur_act(xd, C.split_weight(wp, 2), kern.float().to(dev), sc.float().to(dev), Ci, demod.float().to(dev),
                              noise.float().reshape(-1).to(dev), torch.zeros(1, device=dev), bias.float().to(dev), 2,
                              a_amax=xd.abs().max().reshape(1), a_amax2=sc.float().abs().max().reshape(1).to(dev))
        e = rel_err(y.permute(0, 3, 1, 2), y_x)
        assert e < 1.2e-3, (mag, e)


def test_upconv_fused_rejects_bad_shapes(dev):
    x = torch.zeros(1, 16, 16, 24, device=dev)
    wp = torch.zeros(64, 9, 24, device=dev)
    with pytest.raises(L.WgsError):
        C.upconv_blur_act(x, C.split_weight(wp, 2), blur_kernel().float().to(dev), torch.ones(1, 24, device=dev), 24,
                          torch.ones(1, 64, device=dev), None, None, torch.zeros(64, device=dev), 2)


@pytest.mark.parametrize('B,Ci,Co,H', [(2, 32, 64, 16), (1, 64, 128, 14), (1, 96, 64, 33), (32, 64, 64, 32), (3, 32, 128, 5)])
def test_upconv_fused_split_bf16_is_fp32_class(dev, B, Ci, Co, H):
    """Round 5: the fused kernel in split-bf16 (precision 1: hi + lo planes of BOTH operands, three MFMAs per product) — the arithmetic
    StyleGAN2-256's 32 -> 64 up-sampling layer runs in under the default policy.  fp32-class against the float64 layer, and next to the
    unfused launches (phase GEMMs + blur kernel) of the same arithmetic; both tile shapes (B = 32 at 32 x 32 fills the chip: 8-wave tiles)."""
    torch.manual_seed(Ci * 5 + Co + H)
    x = torch.randn(B, Ci, H, H, dtype=torch.float64)
    w = torch.randn(Co, Ci, 3, 3, dtype=torch.float64) / (Ci * 9) ** 0.5
    sc = (torch.randn(B, Ci) + 1.0).double()
    demod = (torch.rand(B, Co) + 0.5).double()
    noise = torch.randn(2 * H, 2 * H, dtype=torch.float64)
    nw, bias, kern = 0.3, torch.randn(Co, dtype=torch.float64) * 0.2, blur_kernel()
    xs = (x.float() * sc.float()[:, :, None, None]).double()
    y_x = layer_f64(xs, w.float().double(), demod.float().double(), kern.float().double(), noise.float().double(), nw, bias.float().double())
    xd = x.float().permute(0, 2, 3, 1).contiguous().to(dev)
    wp = C.pack_weight(w.float()).to(dev)
    ws = C.split_weight(wp, 1)
    S = sc.float().to(dev).contiguous()
    dm_d, k_d, nz_d = demod.float().to(dev), kern.float().to(dev), noise.float().reshape(-1).to(dev)
    nw_d, b_d = torch.tensor([nw], device=dev), bias.float().to(dev)
    amax = torch.zeros(1, device=dev)
    L.lib().wgs_dev_trace_kernels(1)
    y = C.upconv_blur_act(xd, ws, k_d, S, Ci, dm_d, nz_d, nw_d, b_d, 1, y_amax=amax)
    sym = L.lib().wgs_dev_last_kernel().decode()
    L.lib().wgs_dev_trace_kernels(0)
    assert sym.startswith('upconv_blur_kernel<0, '), sym
    e = rel_err(y.permute(0, 3, 1, 2), y_x)
    assert e < 3e-5, e                                              # split-bf16: ~2^-16 per product
    assert abs(amax.item() - y.abs().max().item()) <= 1e-6 * amax.item()
    t = C.conv_transpose2d_s2(xd, wp, a_scale=S, a_ld=Ci, col_scale=dm_d, w_split=ws, precision=1)
    y2 = torch.empty_like(y)
    L.check(L.lib().wgs_sg2_blur_noise_bias_act(L.ptr(t), L.ptr(k_d), L.ptr(nz_d), L.ptr(nw_d), L.ptr(b_d), L.ptr(y2), None,
                                                B, 2 * H, 2 * H, Co, L.stream()), 'blur_nba')
    assert rel_err(y, y2) < 3e-5
    with pytest.raises(L.WgsError):                                  # one plane only: refused
        C.upconv_blur_act(xd, (ws[0], None), k_d, S, Ci, dm_d, nz_d, nw_d, b_d, 1)
