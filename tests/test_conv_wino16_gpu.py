"""GPU: the split-bf16 F(2,3) x direct conv kernel (csrc/conv_wino_bf16.hip, precision 'bf16x3w') against float64 torch convs and against
the direct split-bf16 kernels — the styled forward conv of models/StyleGAN2/model.py:187-228 (style, demodulation, noise, bias,
leaky-relu epilogue) and its input-gradient form (transposed weights, flipped taps)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_err
from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _small_grids(monkeypatch):
    """The kernel leaves launches of < 200 workgroups to the direct kernels (a dispatch rule, not a coverage limit): the tests lift it."""
    monkeypatch.setenv('WGS_WINO16_MIN_WG', '1')
    L.lib().wgs_dev_reload_flags()
    yield
    monkeypatch.delenv('WGS_WINO16_MIN_WG')
    L.lib().wgs_dev_reload_flags()


TAPS = [(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]


def _supported(x, w, y, **kw):
    d, _ = C._desc(x, w, y, TAPS, y.shape[1], y.shape[2], w_tap_stride=x.shape[3], w_row_stride=9 * x.shape[3], **kw)
    return bool(L.lib().wgs_conv_wino16_supported(ctypes.byref(d)))


@pytest.mark.parametrize('B,H,W,ci,co', [(4, 32, 32, 64, 128), (2, 64, 64, 512, 512), (7, 16, 64, 32, 128), (2, 64, 32, 160, 256), (1, 128, 128, 256, 256)])
def test_wino16_styled_forward_vs_float64(dev, B, H, W, ci, co):
    torch.manual_seed(B * 1000 + ci + co)
    x = torch.randn(B, H, W, ci)
    w = torch.randn(co, ci, 3, 3) / (9 * ci) ** 0.5
    s = torch.randn(B, ci) + 1.0
    dm = torch.rand(B, co) + 0.5
    bias, noise, nw = torch.randn(co), torch.randn(H * W), torch.tensor([0.37])
    xd = (x.double() * s.double()[:, None, None, :]).permute(0, 3, 1, 2)
    ref = F.conv2d(xd, w.double(), padding=1) * dm.double()[:, :, None, None]
    ref = ref + (nw.double() * noise.double()).reshape(1, 1, H, W) + bias.double()[None, :, None, None]
    ref = (F.leaky_relu(ref, 0.2) * 2 ** 0.5).permute(0, 2, 3, 1)
    wp = C.pack_weight(w.to(dev))
    xg = x.to(dev)
    epi = dict(a_scale=s.to(dev), col_scale=dm.to(dev), bias=bias.to(dev), noise=noise.to(dev), noise_w=nw.to(dev), act_slope=0.2, gain=2 ** 0.5)
    y = torch.empty(B, H, W, co, device=dev)
    assert _supported(xg, wp, y, **epi)
    L.lib().wgs_dev_trace_kernels(1)
    ymax = torch.zeros(1, device=dev)
    got = C.conv2d(xg, wp, 3, pad=1, precision=C.BF16W, y_amax=ymax, **epi)
    assert float(ymax) == float(got.abs().max())           # the magnitude scalar the fp16 chains read (exact: a maximum, not a sum)
    sym = L.lib().wgs_dev_last_kernel().decode()
    assert sym == 'wino16_kernel<true, false>', sym
    direct = C.conv2d(xg, wp, 3, pad=1, precision=1, **epi)
    assert 'wino16' not in L.lib().wgs_dev_last_kernel().decode()
    L.lib().wgs_dev_trace_kernels(0)
    e_w, e_d = rel_err(got, ref), rel_err(direct, ref)
    assert e_w < 2e-5, (e_w, e_d)                            # fp32-class: the direct split-bf16 kernels measure the same
    assert rel_err(got, direct) < 3e-5


@pytest.mark.parametrize('B,H,ci,co', [(4, 32, 128, 64), (2, 64, 128, 256), (1, 128, 256, 128)])
def test_wino16_input_gradient_form_vs_float64(dev, B, H, ci, co):
    """dgrad of a 3x3 stride-1 pad-1 conv = the same kernel on dy with the transposed packed weights and flipped taps."""
    torch.manual_seed(7 + ci)
    w = torch.randn(co, ci, 3, 3) / (9 * ci) ** 0.5
    dy = torch.randn(B, H, H, co)
    xd = torch.zeros(B, ci, H, H, dtype=torch.double, requires_grad=True)
    F.conv2d(xd, w.double(), padding=1).backward(dy.double().permute(0, 3, 1, 2))
    ref = xd.grad.permute(0, 2, 3, 1)
    wp = C.pack_weight(w.to(dev))
    wt = C.repack_w_t(wp, co, 9, ci)
    cache = C.SplitCache(wt)
    L.lib().wgs_dev_trace_kernels(1)
    got = C.conv2d_dgrad(dy.to(dev), wt, (H, H), 3, pad=1, precision=C.BF16W, w_split=cache, alpha=1.0)
    sym = L.lib().wgs_dev_last_kernel().decode()
    assert sym == 'wino16_kernel<false, false>', sym
    L.lib().wgs_dev_trace_kernels(0)
    assert rel_err(got, ref) < 2e-5
    assert len(cache.planes) == 1           # U is kept with the weight tensor
    again = C.conv2d_dgrad(dy.to(dev), wt, (H, H), 3, pad=1, precision=C.BF16W, w_split=cache)
    assert torch.equal(got, again) and len(cache.planes) == 1


def test_wino16_declines_what_it_does_not_cover_and_the_direct_kernel_runs(dev):
    torch.manual_seed(3)
    x = torch.randn(8, 16, 16, 64, device=dev)          # 16 wide: not a multiple of the 32-pixel tile
    wp = C.pack_weight(torch.randn(128, 64, 3, 3, device=dev) / 24)
    y = torch.empty(8, 16, 16, 128, device=dev)
    assert not _supported(x, wp, y)
    L.lib().wgs_dev_trace_kernels(1)
    got = C.conv2d(x, wp, 3, pad=1, precision=C.BF16W)
    assert 'wino16' not in L.lib().wgs_dev_last_kernel().decode()
    L.lib().wgs_dev_trace_kernels(0)
    assert torch.equal(got, C.conv2d(x, wp, 3, pad=1, precision=1))       # ... as the direct split-bf16 launch it is
    x3 = torch.randn(8, 32, 32, 64, device=dev)
    assert _supported(x3, wp, torch.empty(8, 32, 32, 128, device=dev))
    assert not _supported(x3, wp, torch.empty(8, 32, 32, 128, device=dev), act=1)


def test_wino16_leaves_small_grids_to_the_direct_kernels(dev, monkeypatch):
    x = torch.randn(2, 32, 32, 64, device=dev)           # 2 x 4 x 1 x 1 = 8 workgroups
    wp = C.pack_weight(torch.randn(128, 64, 3, 3, device=dev) / 24)
    assert _supported(x, wp, torch.empty(2, 32, 32, 128, device=dev))
    monkeypatch.setenv('WGS_WINO16_MIN_WG', '200')
    L.lib().wgs_dev_reload_flags()
    assert not _supported(x, wp, torch.empty(2, 32, 32, 128, device=dev))
    assert _supported(torch.randn(32, 64, 64, 64, device=dev), wp, torch.empty(32, 64, 64, 128, device=dev))


def test_wino16_repeated_launches_are_bit_identical_under_other_traffic(dev):
    """No atomics, fixed summation order: the same launch gives the same bits while other kernels run beside it."""
    torch.manual_seed(11)
    x = torch.randn(8, 64, 64, 128, device=dev)
    wp = C.pack_weight(torch.randn(128, 128, 3, 3, device=dev) / 34)
    s = torch.randn(8, 128, device=dev)
    cache = C.SplitCache(wp)
    first = C.conv2d(x, wp, 3, pad=1, precision=C.BF16W, a_scale=s, w_split=cache)
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=dev)
    for _ in range(10):
        with torch.cuda.stream(side):
            junk = junk * 1.0001
        assert torch.equal(first, C.conv2d(x, wp, 3, pad=1, precision=C.BF16W, a_scale=s, w_split=cache))
    torch.cuda.synchronize()


@pytest.mark.parametrize('keep_y', [True, False])
def test_wino16_torgb_in_the_epilogue(dev, keep_y):
    """wgs_conv_desc.rgb_out with 128 output channels: the three channel sums of models/StyleGAN2/model.py:270-282 from the finished values,
    y stored only when asked for — against the unfused launch + a float64 ToRGB."""
    torch.manual_seed(21)
    B, H, ci, co = 4, 64, 64, 128
    x = torch.randn(B, H, H, ci, device=dev)
    wp = C.pack_weight(torch.randn(co, ci, 3, 3, device=dev) / (9 * ci) ** 0.5)
    s, dm = torch.randn(B, ci, device=dev) + 1.0, torch.rand(B, co, device=dev) + 0.5
    bias, noise, nw = torch.randn(co, device=dev), torch.randn(H * H, device=dev), torch.tensor([0.37], device=dev)
    srgb = torch.randn(B, 300, device=dev)            # a wider style matrix: row stride 300, the layer's slice at column 100
    wrgb = torch.randn(3, co, device=dev)
    epi = dict(a_scale=s, col_scale=dm, bias=bias, noise=noise, noise_w=nw, act_slope=0.2, gain=2 ** 0.5)
    assert C.rgb_wino16_ok(x, wp, **epi)
    plain = C.conv2d(x, wp, 3, pad=1, precision=C.BF16W, **epi)
    rgb = torch.full((B, H, H, 4), 7.0, device=dev)
    L.lib().wgs_dev_trace_kernels(1)
    y = C.conv2d(x, wp, 3, pad=1, precision=C.BF16W, out=None if keep_y else C.NoOutput(B, H, H, co),
                 rgb=dict(out=rgb, s=srgb[:, 100:], ld=300, w=wrgb, scale=0.25), **epi)
    assert L.lib().wgs_dev_last_kernel().decode() == 'wino16_kernel<true, true>'
    L.lib().wgs_dev_trace_kernels(0)
    if keep_y:
        assert torch.equal(y, plain)
    ref = 0.25 * torch.einsum('bhwn,bn,on->bhwo', plain.double(), srgb[:, 100:100 + co].double(), wrgb.double())
    assert rel_err(rgb[..., :3], ref) < 2e-6 and float(rgb[..., 3].abs().max()) == 0.0


@pytest.mark.parametrize('co', [256, 512])
def test_wino16_torgb_partial_sums_per_channel_block(dev, co):
    """rgb_out with 256 / 512 output channels: one 16-byte slot of partial channel sums per 128-channel block and pixel; the finishing launch
    (wgs_sg2_torgb_up_fwd with C = 4 * Co / 128, unit style, tiled identity weight) adds the slots, the bias and the up-sampled skip — against the
    stand-alone ToRGB launch on the layer's output."""
    torch.manual_seed(22)
    B, H, ci = 4, 64, 64
    nb = co // 128
    x = torch.randn(B, H, H, ci, device=dev)
    wp = C.pack_weight(torch.randn(co, ci, 3, 3, device=dev) / (9 * ci) ** 0.5)
    s, dm = torch.randn(B, ci, device=dev) + 1.0, torch.rand(B, co, device=dev) + 0.5
    bias, noise, nw = torch.randn(co, device=dev), torch.randn(H * H, device=dev), torch.tensor([0.37], device=dev)
    srgb = torch.randn(B, 1100, device=dev)
    wrgb = torch.randn(3, co, device=dev)
    epi = dict(a_scale=s, col_scale=dm, bias=bias, noise=noise, noise_w=nw, act_slope=0.2, gain=2 ** 0.5)
    assert C.rgb_wino16_ok(x, wp, **epi)
    plain = C.conv2d(x, wp, 3, pad=1, precision=C.BF16W, **epi)
    rgb = torch.full((B, H, H, 4 * nb), 7.0, device=dev)
    L.lib().wgs_dev_trace_kernels(1)
    y = C.conv2d(x, wp, 3, pad=1, precision=C.BF16W, rgb=dict(out=rgb, s=srgb[:, 100:], ld=1100, w=wrgb, scale=0.25), **epi)
    assert L.lib().wgs_dev_last_kernel().decode() == 'wino16_kernel<true, true>'
    L.lib().wgs_dev_trace_kernels(0)
    assert torch.equal(y, plain)
    part = rgb.view(B, H, H, nb, 4)
    ref = 0.25 * torch.einsum('bhwjn,bjn,ojn->bhwjo', plain.double().view(B, H, H, nb, 128), srgb[:, 100:100 + co].double().view(B, nb, 128),
                              wrgb.double().view(3, nb, 128))
    assert rel_err(part[..., :3], ref) < 2e-6 and float(part[..., 3].abs().max()) == 0.0
    # the finishing launch against the stand-alone ToRGB (+ bias + up-sampled skip) on the stored output
    rb, skip = torch.randn(3, device=dev), torch.randn(B, 3, H // 2, H // 2, device=dev)
    k1 = torch.tensor([1., 3., 3., 1.], device=dev)
    upk = (k1[:, None] * k1[None, :] / 64 * 4).contiguous()
    img_a, img_b = torch.empty(B, 3, H, H, device=dev), torch.empty(B, 3, H, H, device=dev)
    ones, eye = torch.ones(B, 4 * nb, device=dev), torch.eye(3, 4, device=dev).repeat(1, nb).contiguous()
    st = L.stream()
    L.check(L.lib().wgs_sg2_torgb_up_fwd(L.ptr(rgb), L.ptr(ones), 4 * nb, L.ptr(eye), L.ptr(rb), L.ptr(skip), L.ptr(upk), L.ptr(img_a),
                                         B, H, H, 4 * nb, L.c_float(1.0), st), 'finish')
    L.check(L.lib().wgs_sg2_torgb_up_fwd(L.ptr(plain), L.rawptr(srgb[:, 100:]), 1100, L.ptr(wrgb), L.ptr(rb), L.ptr(skip), L.ptr(upk), L.ptr(img_b),
                                         B, H, H, co, L.c_float(0.25), st), 'torgb')
    assert rel_err(img_a, img_b) < 2e-6
