"""GPU: the few-channel 3x3 stride-1 kernel (csrc/conv_halo16.hip, halo3x3_kernel) — the 64- / 32- / 16-channel layers at 256^2 ..
1024^2 of StyleGAN2-1024 (models/StyleGAN2/model.py:297-307) and ProgGAN (models/ProgGAN/model.py:65-95).

  * every 16-bit scheme against a float64 convolution of the SAME rounded operands is not needed to pin it: the GEMM-tiled kernels
    (themselves pinned to fp64, tests/test_conv_gpu.py / test_conv_f16_gpu.py) compute the same products from the same roundings, so the
    two routes must agree to fp32 summation order (2e-6), forward (styled, demodulated, noise, bias, leaky-relu) and input-gradient
    (transposed weights, flipped taps) launches alike; plus the mode's own accuracy against fp64 F.conv2d;
  * the launch really takes the kernel (by symbol), and declines what it does not cover (strides, up-sampling gathers, 1x1, > 64 channels);
  * Cout = 16 (ProgGAN's last block: half-empty column block) and Cin = 16 (one 16-deep chunk per tap)."""
import os

import pytest
import torch
import torch.nn.functional as F

from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C

pytestmark = pytest.mark.gpu
SQRT2 = 2.0 ** 0.5


@pytest.fixture()
def halo_everywhere():
    os.environ['WGS_HALO_MIN_TILES'] = '1'
    L.lib().wgs_dev_reload_flags()
    yield
    os.environ.pop('WGS_HALO_MIN_TILES', None)
    os.environ.pop('WGS_NO_HALO', None)
    L.lib().wgs_dev_reload_flags()


def _route(fn, halo):
    lib = L.lib()
    if halo:
        os.environ.pop('WGS_NO_HALO', None)
    else:
        os.environ['WGS_NO_HALO'] = '1'
    lib.wgs_dev_reload_flags()
    lib.wgs_dev_trace_kernels(1)
    try:
        y = fn()
        sym = lib.wgs_dev_last_kernel().decode()
    finally:
        lib.wgs_dev_trace_kernels(0)
    return y, sym


TOL = {1: 3e-5, 2: 3e-3, 3: 2e-3}


@pytest.mark.parametrize('prec', [1, 2, 3])
@pytest.mark.parametrize('B,Ci,Co,H,W', [(2, 32, 32, 64, 64), (1, 64, 64, 32, 96), (3, 64, 32, 16, 32), (2, 32, 64, 40, 64)])
def test_styled_forward_launch_vs_gemm_tiled_route_and_fp64(dev, halo_everywhere, prec, B, Ci, Co, H, W):
    torch.manual_seed(Ci + Co + H + prec)
    x = torch.randn(B, H, W, Ci, device=dev)
    w = torch.randn(Co, 9, Ci, device=dev) / (9 * Ci) ** 0.5
    S = (torch.randn(B, Ci + 24, device=dev) + 1.0).contiguous()
    s = S[:, 8:]
    dm = torch.rand(B, Co, device=dev) + 0.5
    nz, nw, bias = torch.randn(H * W, device=dev), torch.full((1,), 0.3, device=dev), torch.randn(Co, device=dev) * 0.2
    xm, sm = x.abs().amax().reshape(1), S.abs().amax().reshape(1)
    ws = C.split_weight(w, prec)
    ym = [torch.zeros(1, device=dev), torch.zeros(1, device=dev)]

    def run(i):
        return C.conv2d(x, w, 3, pad=1, a_scale=s, a_ld=S.shape[1], col_scale=dm, noise=nz, noise_w=nw, bias=bias, act_slope=0.2, gain=SQRT2,
                        precision=prec, w_split=ws, a_amax=xm, a_amax2=sm, y_amax=ym[i])
    y1, k1 = _route(lambda: run(0), True)
    y0, k0 = _route(lambda: run(1), False)
    assert k1.startswith('halo3x3_kernel<%d, 32, %d, 3>' % (prec - 1, 64 if Co > 32 else 32)) and not k0.startswith('halo'), (k1, k0)
    assert (y1 - y0).abs().max() <= 2e-6 * y0.abs().max()
    assert ym[0].item() == y1.abs().max().item() and abs(ym[0].item() - ym[1].item()) <= 2e-6 * ym[1].item()
    ref = F.conv2d((x * s[:, None, None, :Ci]).permute(0, 3, 1, 2).double(), w.double().reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2), padding=1)
    ref = F.leaky_relu(ref * dm.double()[:, :, None, None] + 0.3 * nz.double().view(1, 1, H, W) + bias.double()[None, :, None, None], 0.2) * SQRT2
    assert (y1.double().permute(0, 3, 1, 2) - ref).abs().max() <= TOL[prec] * ref.abs().max()


@pytest.mark.parametrize('prec', [1, 2])
@pytest.mark.parametrize('B,Cy,Cx,H', [(2, 32, 32, 64), (2, 64, 64, 32), (1, 16, 16, 64), (2, 32, 16, 32)])
def test_input_gradient_launch_and_narrow_channel_counts(dev, halo_everywhere, prec, B, Cy, Cx, H):
    """dgrad of a conv Cx -> Cy: contraction over Cy (the launch's Ci), Cx output columns; [T, Cx, Cy] weights, flipped taps.  Cy = 16 runs
    16-deep chunks, Cx = 16 a half-empty column block."""
    torch.manual_seed(Cy * 3 + Cx + H + prec)
    dy = torch.randn(B, H, H, Cy, device=dev) * 1e-3
    wp = torch.randn(Cy, 9, Cx, device=dev) / (9 * Cx) ** 0.5            # forward weights [Co = Cy, 9, Ci = Cx]
    wt = C.repack_w_t(wp, Cy, 9, Cx)                                      # [9, Cx, Cy]
    wts = C.split_weight(wt, prec)
    am = dy.abs().amax().reshape(1) * 1.7

    def run():
        return C.conv2d_dgrad(dy, wt, (H, H), 3, pad=1, w_split=wts, a_amax=am, a_bound=1.0, precision=prec)
    g1, k1 = _route(run, True)
    g0, k0 = _route(run, False)
    assert k1.startswith('halo3x3_kernel<%d, %d, %d, 3>' % (prec - 1, 32 if Cy % 32 == 0 else 16, 64 if Cx > 32 else 32)) and not k0.startswith('halo'), (k1, k0)
    assert (g1 - g0).abs().max() <= 2e-6 * g0.abs().max()
    ref = torch.autograd.functional.vjp(lambda xx: F.conv2d(xx, wp.double().reshape(Cy, 3, 3, Cx).permute(0, 3, 1, 2), padding=1),
                                        torch.zeros(B, Cx, H, H, dtype=torch.float64, device=dev), dy.double().permute(0, 3, 1, 2))[1]
    assert (g1.double().permute(0, 3, 1, 2) - ref).abs().max() <= TOL[prec] * ref.abs().max()


@pytest.mark.parametrize('prec', [1, 2])
@pytest.mark.parametrize('B,Ci,Co,H', [(2, 64, 32, 32), (1, 32, 16, 64)])
def test_upsampling_gather_launch(dev, halo_everywhere, prec, B, Ci, Co, H):
    """ProgGAN's nearest-neighbour Upsample + conv (models/ProgGAN/model.py:53-62) as ONE launch: x [B,H,H,Ci] -> y [B,2H,2H,Co], ups = 1."""
    torch.manual_seed(Ci + Co + H + prec)
    x = torch.randn(B, H, H, Ci, device=dev)
    w = torch.randn(Co, 9, Ci, device=dev) / (9 * Ci) ** 0.5
    bias = torch.randn(Co, device=dev) * 0.2
    taps = [(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
    am = x.abs().amax().reshape(1)

    def run():
        y = torch.empty(B, 2 * H, 2 * H, Co, device=dev)
        return C.launch(x, w, y, taps, 2 * H, 2 * H, w_tap_stride=Ci, w_row_stride=9 * Ci, ups=1, alpha=0.7, bias=bias, act_slope=0.2, gain=1.0,
                        w_split=C.split_weight(w, prec), precision=prec, a_amax=am)
    y1, k1 = _route(run, True)
    y0, k0 = _route(run, False)
    assert k1.startswith('halo3x3_kernel<%d, 32, %d, 3>' % (prec - 1, 64 if Co > 32 else 32)) and not k0.startswith('halo'), (k1, k0)
    assert (y1 - y0).abs().max() <= 2e-6 * y0.abs().max()
    xu = F.interpolate(x.permute(0, 3, 1, 2).double(), scale_factor=2, mode='nearest')
    ref = F.leaky_relu(F.conv2d(xu, w.double().reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2), padding=1) * 0.7 + bias.double()[None, :, None, None], 0.2)
    assert (y1.double().permute(0, 3, 1, 2) - ref).abs().max() <= TOL[prec] * ref.abs().max()


def test_trained_weights_without_planes_give_the_same_bits(dev, halo_everywhere):
    """The Reconstructor's convs carry no pre-split weight planes (the weights change every step): the pre-pass splits the fp32 weights
    itself, with the roundings of wgs_split_bf16 / wgs_split_f16."""
    torch.manual_seed(1)
    x = torch.randn(2, 32, 64, 64, device=dev)
    w = torch.randn(64, 9, 64, device=dev) * 0.04
    for prec in (1, 2, 3):
        a = C.conv2d(x, w, 3, pad=1, precision=prec, w_split=C.split_weight(w, prec))
        b = C.conv2d(x, w, 3, pad=1, precision=prec)
        assert torch.equal(a, b)


def test_shapes_the_kernel_declines(dev, halo_everywhere):
    lib = L.lib()
    w = torch.randn(32, 9, 32, device=dev) * 0.05
    ws = C.split_weight(w, 1)
    x = torch.randn(1, 64, 64, 32, device=dev)
    lib.wgs_dev_trace_kernels(1)
    try:
        C.conv2d(x, w, 3, stride=2, pad=1, precision=1, w_split=ws)                                   # strided
        assert not lib.wgs_dev_last_kernel().decode().startswith('halo')
        C.conv2d(x[:, :60].contiguous(), w, 3, pad=1, precision=1, w_split=ws)                        # height not a multiple of 8
        assert not lib.wgs_dev_last_kernel().decode().startswith('halo')
        C.conv2d(x, w, 3, pad=1, precision=1)                                                         # no pre-split planes: split from the fp32 weights in the pre-pass
        assert lib.wgs_dev_last_kernel().decode().startswith('halo3x3_kernel<0, 32, 32, 3>')
        w1 = torch.randn(32, 1, 32, device=dev)
        C.conv2d(x, w1, 1, precision=1, w_split=C.split_weight(w1, 1))                                # 1x1
        assert not lib.wgs_dev_last_kernel().decode().startswith('halo')
        w128 = torch.randn(128, 9, 32, device=dev) * 0.05
        C.conv2d(x, w128, 3, pad=1, precision=1, w_split=C.split_weight(w128, 1))                     # 128 output channels: the patch / GEMM kernels
        assert not lib.wgs_dev_last_kernel().decode().startswith('halo')
        C.conv2d(x, w, 3, pad=1, precision=1, w_split=ws)
        assert lib.wgs_dev_last_kernel().decode().startswith('halo3x3_kernel<0, 32, 32, 3>')
    finally:
        lib.wgs_dev_trace_kernels(0)


@pytest.mark.parametrize('Ci,Co,H,prec,with_y', [(64, 64, 256, 'f16x2', True), (32, 32, 256, 'f16x2', False), (64, 32, 256, 'f16', True), (32, 64, 256, 'bf16x3', True)])
def test_torgb_in_the_few_channel_kernel_epilogue(dev, Ci, Co, H, prec, with_y):
    """wgs_conv_desc.rgb_out without an fp16 operand plane: ToRGB's channel sums from the epilogue of the few-channel kernel (StyleGAN2-1024's
    64- / 32-channel layers at 512^2 / 1024^2) against wgs_sg2_torgb_fwd on the stored output; y bit-identical to the launch without it."""
    import torch
    from warpedganspace_amd import _lib as L, conv as C
    torch.manual_seed(Ci + Co + H)
    lib, st = L.lib(), L.stream()
    B = 4
    x = torch.randn(B, H, H, Ci, device=dev)
    S = (torch.randn(B, Ci + Co + 20, device=dev) + 1.0).contiguous()
    s_in, s_rgb = S[:, 4:], S[:, Ci + 10:]
    w = torch.randn(Co, 9, Ci, device=dev) / (9 * Ci) ** 0.5
    m = C.precision_code(prec)
    demod = torch.rand(B, Co, device=dev) + 0.5
    noise, nw, bias = torch.randn(H * H, device=dev), torch.full((1,), 0.3, device=dev), torch.randn(Co, device=dev) * 0.2
    am_in = (x.abs().amax()).reshape(1)
    smax = s_in.abs().amax().reshape(1)
    epi = dict(a_scale=s_in, a_ld=S.shape[1], col_scale=demod, noise=noise, noise_w=nw, bias=bias, act_slope=0.2, gain=2 ** 0.5, w_split=C.SplitCache(w),
               **(dict(a_amax=am_in, a_amax2=smax) if m in (2, 3) else {}))
    assert C.rgb_halo_ok(x, w, m, **epi)
    ref = C.conv2d(x, w, 3, pad=1, precision=m, **epi)
    w_rgb = torch.randn(3, Co, device=dev).contiguous()
    rgbp = torch.full((B, H, H, 4), float('nan'), device=dev)
    am = torch.zeros(1, device=dev)
    out = torch.empty(B, H, H, Co, device=dev) if with_y else C.NoOutput(B, H, H, Co)
    lib.wgs_dev_trace_kernels(1)
    try:
        got = C.conv2d(x, w, 3, pad=1, precision=m, out=out, y_amax=am, rgb=dict(out=rgbp, s=s_rgb, ld=S.shape[1], w=w_rgb, scale=0.37), **epi)
        sym = lib.wgs_dev_last_kernel().decode()
    finally:
        lib.wgs_dev_trace_kernels(0)
    assert sym.startswith('halo3x3_kernel') and sym.endswith('true>'), sym
    if with_y:
        assert torch.equal(got, ref)
    assert am.item() == ref.abs().max().item()
    want = torch.empty(B, 3, H * H, device=dev)
    s_c, b0 = s_rgb[:, :Co].contiguous(), torch.zeros(3, device=dev)
    L.check(lib.wgs_sg2_torgb_fwd(L.ptr(ref), L.ptr(s_c), L.ptr(w_rgb), L.ptr(b0), None, L.ptr(want), B, H * H, Co, L.c_float(0.37), st), 'torgb')
    gotc = rgbp.view(B, H * H, 4)
    assert float(gotc[..., 3].abs().max()) == 0.0
    assert (gotc[..., :3].permute(0, 2, 1) - want).abs().max() <= 3e-6 * want.abs().max()
    # a launch the kernel does not take may not carry rgb_out
    x2, w2 = torch.randn(2, 16, 16, 64, device=dev), torch.randn(64, 9, 64, device=dev)
    assert not C.rgb_halo_ok(x2, w2, m, w_split=C.SplitCache(w2))
    with pytest.raises(L.WgsError):
        C.conv2d(x2, w2, 3, pad=1, precision=m, w_split=C.SplitCache(w2), rgb=dict(out=torch.empty(2, 16, 16, 4, device=dev), s=s_rgb, ld=S.shape[1], w=w_rgb, scale=1.0))
