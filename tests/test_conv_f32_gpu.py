"""GPU: the slot-interleaved exact-fp32 conv kernel (csrc/conv_igemm_f32.hip, scheme 4 of the register-staged template) against
torch-CPU fp64 convolutions and against the plain fp32 kernel it replaces (conv_igemm.hip, WGS_F32_OLD): same arithmetic, so the
two agree to the summation order inside a 32-deep chunk (~1e-6), on every gather form of the hot path."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_err
from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def _symbol_of(fn):
    lib = L.lib()
    lib.wgs_dev_trace_kernels(1)
    try:
        fn()
        return lib.wgs_dev_last_kernel().decode()
    finally:
        lib.wgs_dev_trace_kernels(0)


# B, Ci, Co, H, k, stride, pad — tile remainders in M and Co, several samples per tile, split-K sized maps, big maps
CASES = [(2, 32, 64, 9, 3, 1, 1), (3, 64, 128, 8, 3, 1, 1), (2, 128, 160, 6, 3, 1, 1), (2, 64, 128, 9, 3, 2, 1), (2, 64, 128, 8, 1, 2, 0),
         (1, 512, 512, 4, 3, 1, 1), (5, 32, 32, 3, 3, 1, 1), (2, 128, 128, 32, 3, 1, 1), (4, 256, 96, 16, 3, 1, 1), (2, 64, 24, 12, 4, 1, 2), (8, 256, 256, 40, 3, 1, 1)]


@pytest.mark.parametrize('B,Ci,Co,H,k,s,p', CASES)
@pytest.mark.parametrize('small', [False, True])
def test_f32_conv_fwd_and_dgrad_vs_fp64_and_old_kernel(dev, dev_flags, B, Ci, Co, H, k, s, p, small):
    torch.manual_seed(B * 1000 + Ci + Co + H)
    x = torch.randn(B, Ci, H, H, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Co, Ci, k, k, dtype=torch.float64) / (Ci * k * k) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, stride=s, padding=p)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    xd = nhwc(x.detach().float()).to(dev)
    wp = C.pack_weight(w.detach().float()).to(dev)
    wt = C.repack_w_t(wp, Co, k * k, Ci)
    gd = nhwc(g.float()).to(dev)
    if small:
        dev_flags(WGS_F32_SMALL='1')
    sym = _symbol_of(lambda: C.conv2d(xd, wp, k, stride=s, pad=p, precision=0))
    assert sym.startswith('igemm_nt16_kernel<4,'), sym           # the launch really went through the new kernel
    yd = C.conv2d(xd, wp, k, stride=s, pad=p, precision=0)
    dx = C.conv2d_dgrad(gd, wt, (H, H), k, stride=s, pad=p, precision=0)
    assert rel_err(nchw(yd), y.detach()) < 1e-5
    assert rel_err(nchw(dx), x.grad) < 1e-5
    dev_flags(WGS_F32_OLD='1')
    assert _symbol_of(lambda: C.conv2d(xd, wp, k, stride=s, pad=p, precision=0)).startswith('igemm_nt_kernel<')
    yo = C.conv2d(xd, wp, k, stride=s, pad=p, precision=0)
    dxo = C.conv2d_dgrad(gd, wt, (H, H), k, stride=s, pad=p, precision=0)
    assert rel_err(yd, yo) < 1e-5 and rel_err(dx, dxo) < 1e-5          # same products, another summation order inside a chunk


@pytest.mark.parametrize('B,Ci,Co,H', [(2, 64, 32, 5), (3, 128, 128, 8), (2, 256, 128, 33), (32, 128, 128, 16)])
@pytest.mark.parametrize('small', [False, True])
def test_f32_styled_transposed_conv_and_its_dgrad(dev, dev_flags, B, Ci, Co, H, small):
    """StyleGAN2's up-sampling conv (4 sub-pixel phases, style on the activation, demodulation on the columns) and the stride-2
    gradient conv behind it."""
    torch.manual_seed(77 + Ci + H)
    x = torch.randn(B, Ci, H, H)
    w = torch.randn(Ci, Co, 3, 3) / (Ci * 9) ** 0.5                    # conv_transpose2d weight [in, out, k, k]
    s = torch.randn(B, Ci) + 1.0
    dm = torch.rand(B, Co) + 0.5
    ref = F.conv_transpose2d((x * s[:, :, None, None]).double(), w.double(), stride=2) * dm[:, :, None, None].double()
    wp = C.pack_weight(w.permute(1, 0, 2, 3).contiguous()).to(dev)      # [Co, 9, Ci]
    if small:
        dev_flags(WGS_F32_SMALL='1')
    t = C.conv_transpose2d_s2(nhwc(x).to(dev), wp, a_scale=s.to(dev), col_scale=dm.to(dev), precision=0)
    assert rel_err(nchw(t), ref) < 1e-5
    g = torch.randn(B, Co, 2 * H + 1, 2 * H + 1)
    gref = F.conv2d(g.double(), w.double(), stride=2)                   # gradient of conv_transpose2d w.r.t. its input
    wt = C.repack_w_t(wp, Co, 9, Ci)
    dx = C.conv_transpose2d_s2_dgrad(nhwc(g).to(dev), wt, precision=0)
    assert rel_err(nchw(dx), gref) < 1e-5
    dev_flags(WGS_F32_OLD='1')
    to = C.conv_transpose2d_s2(nhwc(x).to(dev), wp, a_scale=s.to(dev), col_scale=dm.to(dev), precision=0)
    assert rel_err(t, to) < 1e-5


def test_f32_upsampled_gather_epilogue_and_addend(dev, dev_flags):
    """ProgGAN / SNGAN / BigGAN block form: nearest 2x up-sampling folded into the gather, alpha, bias, leaky-relu, a low-resolution
    addend, and the tanh output epilogue."""
    torch.manual_seed(9)
    B, Ci, Co, H = 3, 64, 96, 6
    x = torch.randn(B, Ci, H, H)
    w = torch.randn(Co, Ci, 3, 3) / (Ci * 9) ** 0.5
    bias, add = torch.randn(Co), torch.randn(B, Co, H, H)
    up = F.interpolate(x.double(), scale_factor=2, mode='nearest')
    ref = F.leaky_relu(0.7 * F.conv2d(up, w.double(), padding=1) + bias.double()[None, :, None, None]
                       + F.interpolate(add.double(), scale_factor=2, mode='nearest'), 0.2)
    taps = [(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
    y = torch.empty(B, 2 * H, 2 * H, Co, device=dev)
    args = dict(w_tap_stride=Ci, w_row_stride=9 * Ci, ups=1, alpha=0.7, bias=bias.to(dev), act_slope=0.2, gain=1.0, addend=nhwc(add).to(dev),
                add_ups=1, precision=0)
    assert _symbol_of(lambda: C.launch(nhwc(x).to(dev), C.pack_weight(w).to(dev), y, taps, 2 * H, 2 * H, **args)).startswith('igemm_nt16_kernel<4,')
    assert rel_err(nchw(y), ref) < 1e-5
    y2 = torch.empty(B, 2 * H, 2 * H, Co, device=dev)
    C.launch(nhwc(x).to(dev), C.pack_weight(w).to(dev), y2, taps, 2 * H, 2 * H, w_tap_stride=Ci, w_row_stride=9 * Ci, ups=1, act=1, precision=0)
    assert rel_err(nchw(y2), torch.tanh(F.conv2d(up, w.double(), padding=1))) < 1e-5


def test_f32_falls_back_where_the_template_does_not_reach(dev):
    """Ci % 32 != 0 (ResNet conv1: 8 padded channels; LeNet) and more than 16 taps: the plain fp32 kernel keeps those launches."""
    x = torch.randn(2, 20, 20, 8, device=dev)
    w = torch.randn(64, 49, 8, device=dev)
    assert _symbol_of(lambda: C.conv2d(x, w, 7, stride=2, pad=3, precision=0)).startswith('igemm_nt_kernel<')
    x = torch.randn(2, 12, 12, 64, device=dev)
    w = torch.randn(24, 25, 64, device=dev)
    assert _symbol_of(lambda: C.conv2d(x, w, 5, stride=1, pad=2, precision=0)).startswith('igemm_nt_kernel<')


def test_conv_beyond_2gib_is_split_over_samples(dev):
    """A [B, H, W, C] operand larger than 2 GiB (ProgGAN-1024's feature maps at batch 32) is issued as consecutive launches over
    sample ranges, each through the fast template: same result per sample as that sample on its own, with style / demodulation rows,
    bias and a low-resolution addend following their sample."""
    B, H, Ci, Co = 3, 1536, 96, 32                      # x: 3 x 1536^2 x 96 x 4 B = 2.7 GB; y 0.9 GB
    torch.manual_seed(3)
    x = torch.randn(B, H, H, Ci, device=dev)
    w = torch.randn(Co, 9, Ci, device=dev) / (9 * Ci) ** 0.5
    s, dm, bias = torch.randn(B, Ci, device=dev) + 1.0, torch.rand(B, Co, device=dev) + 0.5, torch.randn(Co, device=dev)
    kw = dict(a_scale=s, col_scale=dm, bias=bias, act_slope=0.2, gain=1.3)
    for prec in (0, 1):
        sym = _symbol_of(lambda: C.conv2d(x, w, 3, pad=1, precision=prec, **kw))
        assert sym.startswith('igemm_nt16_kernel<'), sym
        y = C.conv2d(x, w, 3, pad=1, precision=prec, **kw)
        for b in (0, B - 1):
            yb = C.conv2d(x[b:b + 1].contiguous(), w, 3, pad=1, precision=prec, a_scale=s[b:b + 1].contiguous(), col_scale=dm[b:b + 1].contiguous(),
                          bias=bias, act_slope=0.2, gain=1.3)
            assert torch.equal(y[b:b + 1], yb), (prec, b)
        del y, yb


@pytest.mark.parametrize('prec', [0, 1, 2])
@pytest.mark.parametrize('B,Ci,Co,H,k,ups', [(2, 16, 16, 21, 3, 0), (3, 16, 32, 16, 3, 0), (2, 16, 8, 19, 2, 0), (2, 16, 16, 9, 3, 1), (1, 16, 64, 12, 4, 0)])
def test_sixteen_input_channels_pair_their_taps(dev, B, Ci, Co, H, k, ups, prec):
    """Ci = 16 (ProgGAN's 512^2 / 1024^2 layers): the template stages two taps x 16 channels per 32-deep chunk (an odd last tap
    pairs with zeros); every arithmetic scheme, plain and up-sampled gathers, against a float64 convolution."""
    torch.manual_seed(B + Co + H + k)
    x = torch.randn(B, Ci, H, H)
    w = torch.randn(Co, Ci, k, k) / (Ci * k * k) ** 0.5
    bias = torch.randn(Co)
    pad = k // 2
    xin = F.interpolate(x.double(), scale_factor=2, mode='nearest') if ups else x.double()
    ref = F.leaky_relu(0.9 * F.conv2d(xin, w.double(), padding=pad) + bias.double()[None, :, None, None], 0.2)
    Ho = ref.shape[2]
    taps = [(ky - pad, kx - pad, ky * k + kx) for ky in range(k) for kx in range(k)]
    y = torch.empty(B, Ho, Ho, Co, device=dev)
    xd, wd = nhwc(x).to(dev), C.pack_weight(w).to(dev)
    amax = xd.abs().amax().reshape(1)
    kw = dict(w_tap_stride=Ci, w_row_stride=k * k * Ci, ups=ups, alpha=0.9, bias=bias.to(dev), act_slope=0.2, gain=1.0, precision=prec,
              a_amax=amax if prec >= 2 else None)
    sym = _symbol_of(lambda: C.launch(xd, wd, y, taps, Ho, Ho, **kw))
    assert sym.startswith('igemm_nt16_kernel<%d,' % (4 if prec == 0 else prec - 1)) and ', 3, ' in sym, sym
    assert rel_err(nchw(y), ref) < (1e-5 if prec == 0 else 1e-4 if prec == 1 else 2e-3)
