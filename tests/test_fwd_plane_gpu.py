"""GPU: forward fp16 activation planes — the fused up-sampling kernel writes the operand plane of the
stride-1 conv behind it (wgs_upconv_desc.y_f16: that conv's style vector and power-of-two scale folded in), and the conv stages
the plane as it is through the patch kernel's XF16 form (wgs_conv_desc.x_f16).  Replaces, between the two kernels, the fp32 tensor
the reference moves (models/StyleGAN2/model.py:187-228 modulation of the NEXT layer, op/fused_bias_act_kernel.cu:18-49 the store).

  * the plane holds exactly f16_rn(fl32(y * s_next) * 2^k), k from the a-priori bound the kernel publishes; y itself is unchanged;
  * a pass that keeps nothing (y = NULL) writes the same plane;
  * the conv fed with the plane returns the SAME BITS as the conv fed with the fp32 tensor + style vector under the same scale,
    through the patch kernel (Cout < 512) and through the LDS-DMA kernel;
  * the a-priori bound is a bound (>= the true maximum) and not absurdly loose;
  * a generator forward / backward with the route on and off: same image bits (scales differ by powers of two only), same gradient."""
import math

import pytest
import torch

from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C

pytestmark = pytest.mark.gpu
SQRT2 = 2.0 ** 0.5


def blur_kernel(dev):
    k = torch.tensor([1.0, 3.0, 3.0, 1.0])
    k2 = k[:, None] * k[None, :]
    return (k2 / k2.sum() * 4.0).to(dev).contiguous()


def _layer(dev, B, Ci, Co, H, seed, xmag=1.0):
    torch.manual_seed(seed)
    x = (torch.randn(B, H, H, Ci, device=dev) * xmag).contiguous()
    w = torch.randn(Co, 9, Ci, device=dev) / (9 * Ci) ** 0.5
    sumC = Ci + Co + 40
    S = (torch.randn(B, sumC, device=dev) + 1.0).contiguous()
    wsq = (w * w).sum(1)                                                      # [Co, Ci]
    demod = torch.rsqrt((S[:, :Ci] ** 2) @ wsq.t() + 1e-8).contiguous()      # the layer's real demodulation vector: the bound relies on it
    noise, nw, bias = torch.randn(4 * H * H, device=dev), torch.full((1,), 0.3, device=dev), torch.randn(Co, device=dev) * 0.2
    xmax = x.abs().amax().reshape(1).contiguous()
    smax = S.abs().amax().reshape(1).contiguous()
    mul = SQRT2 * 4.0 * 2.0 * math.sqrt(Ci)
    add = SQRT2 * (0.3 * float(noise.abs().max()) + float(bias.abs().max()))
    return dict(x=x, w=w, ws=C.split_weight(w, 3), S=S, sumC=sumC, demod=demod, noise=noise, nw=nw, bias=bias, xmax=xmax, smax=smax, mul=mul, add=add,
                s_next=S[:, Ci + 8:])


def _k_of(a):
    k = 0
    while a * 2.0 ** k < 2048.0:
        k += 1
    while a * 2.0 ** k >= 4096.0:
        k -= 1
    return k


@pytest.mark.parametrize('prec', [2, 3])
@pytest.mark.parametrize('B,Ci,Co,H,xmag', [(2, 64, 128, 32, 1.0), (1, 32, 64, 20, 1e-5), (3, 96, 128, 16, 3e4), (2, 64, 64, 33, 1.0)])
def test_plane_bits_bound_and_unchanged_output(dev, prec, B, Ci, Co, H, xmag):
    t = _layer(dev, B, Ci, Co, H, 11 * Ci + H + prec, xmag)
    kern = blur_kernel(dev)
    args = (t['x'], t['ws'], kern, t['S'], t['sumC'], t['demod'], t['noise'], t['nw'], t['bias'], prec)
    kw = dict(a_amax=t['xmax'], a_amax2=t['smax'])
    y0 = C.upconv_blur_act(*args, **kw)
    pl = dict(scale=t['s_next'], ld=t['sumC'], mul=t['mul'], add=t['add'])
    y1, plane, bound = C.upconv_blur_act(*args, plane=dict(pl, keep_y=True), **kw)
    assert torch.equal(y0, y1)
    yn, plane2, bound2 = C.upconv_blur_act(*args, plane=dict(pl, keep_y=False), **kw)
    assert yn is None and torch.equal(plane, plane2) and torch.equal(bound, bound2)
    want_bound = (t['xmax'] * t['mul'] + t['add']) * t['smax']
    assert abs(bound.item() - want_bound.item()) <= 1e-6 * want_bound.item()
    ys = (y0 * t['s_next'][:, None, None, :Co])                              # fl32(y * s)
    true_max = ys.abs().max().item()
    assert true_max <= bound.item() <= 4096.0 * true_max, (true_max, bound.item())     # a bound, and inside fp16's exponent head-room
    k = _k_of(bound.item())
    assert torch.equal(plane, (ys * 2.0 ** k).half().view(torch.int16))


@pytest.mark.parametrize('B,C1,C2,H,route', [(32, 128, 128, 64, 'patch'), (8, 256, 256, 64, 'patch'), (32, 128, 128, 64, 'dma'), (4, 64, 256, 64, 'auto')])
def test_conv_fed_with_the_plane_returns_the_same_bits(dev, B, C1, C2, H, route):
    """y [B,H,H,C1] -> 3x3 stride-1 conv to C2 channels, plain fp16: plane route vs fp32 tensor + style vector under the same scale."""
    torch.manual_seed(C1 + C2 + H)
    lib = L.lib()
    y = torch.randn(B, H, H, C1, device=dev)
    S = (torch.randn(B, C1 + 30, device=dev) + 1.0).contiguous()
    s_next = S[:, 6:]
    w = torch.randn(C2, 9, C1, device=dev) / (9 * C1) ** 0.5
    ws = C.split_weight(w, 2)
    demod = torch.rand(B, C2, device=dev) + 0.5
    noise, nw, bias = torch.randn(H * H, device=dev), torch.full((1,), 0.3, device=dev), torch.randn(C2, device=dev) * 0.2
    bound = ((y * s_next[:, None, None, :C1]).abs().amax() * 37.0).reshape(1).contiguous()        # any over-estimate
    k = _k_of(bound.item())
    plane = ((y * s_next[:, None, None, :C1]) * 2.0 ** k).half().view(torch.int16).contiguous()
    epi = dict(col_scale=demod, noise=noise, noise_w=nw, bias=bias, act_slope=0.2, gain=SQRT2, w_split=ws, precision=2)
    ref = C.conv2d(y, w, 3, pad=1, a_scale=s_next, a_ld=S.shape[1], a_amax=bound, a_bound=1.0, **epi)
    import os
    if route != 'auto':
        os.environ['WGS_PLANE_PATCH_MAX_CO'] = '100000' if route == 'patch' else '0'
        lib.wgs_dev_reload_flags()
    try:
        lib.wgs_dev_trace_kernels(1)
        got = C.conv2d(plane, w, 3, pad=1, a_amax=bound, a_bound=1.0, x_f16=True, out=torch.empty_like(ref), **epi)
        sym = lib.wgs_dev_last_kernel().decode()
    finally:
        lib.wgs_dev_trace_kernels(0)
        os.environ.pop('WGS_PLANE_PATCH_MAX_CO', None)
        lib.wgs_dev_reload_flags()
    want = {'patch': 'igemm_patch_kernel', 'dma': 'igemm_dma16_kernel', 'auto': 'igemm_dma16_kernel'}[route]       # auto: Cout >= 256 -> LDS-DMA
    if route == 'patch' and C2 == 128:
        want = 'patch_dma_kernel'            # the 128 x 128 tile of a plane in plain fp16: every operand by LDS-DMA (conv_patch_dma.hip)
    assert sym.startswith(want) and ((', true>' in sym) if want == 'igemm_patch_kernel' else True), sym
    if route == 'patch':
        assert torch.equal(got, ref)              # same kernel family, same tile, same MFMA order: same bits
    assert (got - ref).abs().max() <= 2e-6 * ref.abs().max()
    full = torch.nn.functional.conv2d((y * s_next[:, None, None, :C1]).permute(0, 3, 1, 2).double(),
                                      w.double().reshape(C2, 3, 3, C1).permute(0, 3, 1, 2), padding=1)
    full = torch.nn.functional.leaky_relu(full * demod.double()[:, :, None, None] + 0.3 * noise.double().view(1, 1, H, H) + bias.double()[None, :, None, None], 0.2) * SQRT2
    assert (got.double().permute(0, 3, 1, 2) - full).abs().max() <= 2e-3 * full.abs().max()


@pytest.mark.parametrize('size,prec', [(256, 'mixed'), (128, 'f16')])
def test_generator_same_image_and_gradient_with_and_without_forward_planes(dev, monkeypatch, size, prec):
    from tests import golden_inputs as GI
    from warpedganspace_amd.stylegan2 import Generator
    torch.manual_seed(0)
    G = Generator(size, 512, 8)
    G.load_state_dict(GI.fill_state_dict(G.state_dict(), 977))
    G = G.to(dev).eval()
    G.precision = prec
    B = 32 if size == 128 else 16
    z = torch.randn(B, 512, device=dev)
    imgs, grads, nograd = [], [], []
    lib = L.lib()
    for on in (True, False):
        monkeypatch.setattr(C, 'FWD_PLANE', on)
        with torch.no_grad():
            nograd.append(G([z], precision=prec)[0].clone())        # the pass that keeps nothing: the up-convs write only the planes
        zz = z.clone().requires_grad_(True)
        img = G([zz], precision=prec)[0]
        gi = torch.linspace(-1, 1, img.numel(), device=dev).view_as(img)
        img.backward(gi)
        imgs.append(img.detach().clone()); grads.append(zz.grad.clone())
    assert torch.isfinite(imgs[0]).all() and torch.isfinite(grads[0]).all()
    # the pass that keeps nothing and the pass that saves for the backward take the same route: same bits
    assert torch.equal(nograd[0], imgs[0]) and torch.equal(nograd[1], imgs[1])
    # On vs off: every operand gets the same fp16 rounding unless it lies below the looser (a-priori) scale's normal range; those
    # values (activations next to a leaky-relu zero crossing) move the consuming conv's fp32 sums by ~1 ulp in ~1 % of its outputs, and
    # a 1-ulp fp32 difference flips the fp16 rounding of the NEXT layer's operand once in 2^13: measured 2e-5 of the image maximum
    # (the fp16 mode's own error is 5e-4, the gate 1e-3).  Bit-equality of the kernels themselves is asserted above.
    e_img = float((imgs[0] - imgs[1]).abs().max() / imgs[1].abs().max())
    e_g = float((grads[0] - grads[1]).abs().max() / grads[1].abs().max())
    print('forward planes on vs off, StyleGAN2-%d %s: image %.2e, gradient %.2e' % (size, prec, e_img, e_g))
    assert e_img <= 1e-4
    assert e_g <= 1e-3


def test_the_route_is_taken_in_the_default_policy(dev):
    """StyleGAN2-256 under the default 'mixed' table at the training batch: both plain-fp16 stride-1 convs (128^2, 256^2) read a producer-written plane."""
    from warpedganspace_amd.gan_load import build_stylegan2
    torch.manual_seed(0)
    G = build_stylegan2(None, resolution=256).to(dev).eval()
    z = torch.randn(32, 512, device=dev)
    C.PROFILE = []
    L.lib().wgs_dev_trace_kernels(1)
    try:
        with torch.no_grad():
            G(z, precision='mixed')
        torch.cuda.synchronize()
        syms = [r[4] for r in C.PROFILE if r[0] and ' 9 taps' in r[0] and (r[0].startswith('conv f16 128->128 @256') or r[0].startswith('conv f16 256->256 @128'))]
    finally:
        L.lib().wgs_dev_trace_kernels(0)
        C.PROFILE = None
    # the 128-column launch through the all-DMA patch kernel, the 256-column one through the LDS-DMA kernel: neither stages fp32
    assert len(syms) == 2 and sorted(s.split('<')[0] for s in syms) == ['igemm_dma16_kernel', 'patch_dma_kernel'], syms


@pytest.mark.parametrize('B,Ci,Co,H,with_y', [(13, 128, 128, 64, True), (13, 128, 128, 64, False), (14, 64, 128, 64, True), (8, 128, 128, 64, True), (8, 128, 128, 64, False),
                                               (16, 256, 256, 64, True), (13, 128, 256, 64, False)])
def test_torgb_in_the_conv_epilogue(dev, B, Ci, Co, H, with_y):
    """wgs_conv_desc.rgb_out: ToRGB's channel sums (models/StyleGAN2/model.py:270-282) from the epilogue of the 128-channel conv that
    produces its input — against wgs_sg2_torgb_fwd on that conv's stored output (fp32 summation order), with and without storing y;
    y itself is bit-identical to the launch without the fusion."""
    import os
    torch.manual_seed(B + Ci + H)
    lib, st = L.lib(), L.stream()
    os.environ['WGS_HALO_MIN_TILES'] = '1000000'; lib.wgs_dev_reload_flags()
    try:
        yin = torch.randn(B, H, H, Ci, device=dev)
        S = (torch.randn(B, Ci + Co + 20, device=dev) + 1.0).contiguous()
        s_in, s_rgb = S[:, 4:], S[:, Ci + 10:]
        w = torch.randn(Co, 9, Ci, device=dev) / (9 * Ci) ** 0.5
        ws = C.split_weight(w, 2)
        demod = torch.rand(B, Co, device=dev) + 0.5
        noise, nw, bias = torch.randn(H * H, device=dev), torch.full((1,), 0.3, device=dev), torch.randn(Co, device=dev) * 0.2
        bound = ((yin * s_in[:, None, None, :Ci]).abs().amax() * 3.0).reshape(1).contiguous()
        k = _k_of(bound.item())
        plane = ((yin * s_in[:, None, None, :Ci]) * 2.0 ** k).half().view(torch.int16).contiguous()
        w_rgb = torch.randn(3, Co, device=dev).contiguous()
        epi = dict(col_scale=demod, noise=noise, noise_w=nw, bias=bias, act_slope=0.2, gain=SQRT2, w_split=ws, precision=2, a_amax=bound, a_bound=1.0, x_f16=True)
        ref = C.conv2d(plane, w, 3, pad=1, out=torch.empty(B, H, H, Co, device=dev), **epi)
        rgbp = torch.full((B, H, H, 4), float('nan'), device=dev)
        am = torch.zeros(1, device=dev)
        out = torch.empty(B, H, H, Co, device=dev) if with_y else C.NoOutput(B, H, H, Co)
        lib.wgs_dev_trace_kernels(1)
        got = C.conv2d(plane, w, 3, pad=1, out=out, y_amax=am, rgb=dict(out=rgbp, s=s_rgb, ld=S.shape[1], w=w_rgb, scale=0.37), **epi)
        sym = lib.wgs_dev_last_kernel().decode()
        lib.wgs_dev_trace_kernels(0)
        # Cout = 128: the all-DMA patch kernel when its 256-pixel tiles fill the chip (>= 200 of them), else the register-staged one
        want = 'igemm_dma16_kernel<1, 256, 256, 2, 4, true>' if Co != 128 else \
            ('patch_dma_kernel<256, 128, true>' if B * (H // 16) ** 2 >= 200 else 'igemm_patch_kernel<1, 128, 128, 2, 2, 1, 0, true, true>')
        assert sym.startswith(want), sym
        if with_y:
            assert torch.equal(got, ref)
        assert am.item() == ref.abs().max().item()
        want = torch.empty(B, 3, H * H, device=dev)
        s_c, b0 = s_rgb[:, :Co].contiguous(), torch.zeros(3, device=dev)       # (named: a temporary dies — and its block is re-used — before the launch)
        L.check(lib.wgs_sg2_torgb_fwd(L.ptr(ref), L.ptr(s_c), L.ptr(w_rgb), L.ptr(b0), None, L.ptr(want), B, H * H, Co, L.c_float(0.37), st), 'torgb')
        gotc = rgbp.view(B, H * H, 4)
        assert float(gotc[..., 3].abs().max()) == 0.0
        assert (gotc[..., :3].permute(0, 2, 1) - want).abs().max() <= 3e-6 * want.abs().max()
        f64 = torch.einsum('bpc,bc,oc->bop', ref.double().view(B, H * H, Co), s_rgb[:, :Co].double(), w_rgb.double()) * 0.37
        assert (gotc[..., :3].permute(0, 2, 1).double() - f64).abs().max() <= 3e-6 * f64.abs().max()
    finally:
        os.environ.pop('WGS_HALO_MIN_TILES', None); lib.wgs_dev_reload_flags()


def test_rgb_epilogue_rejects_other_shapes(dev):
    x = torch.zeros(1, 16, 16, 64, device=dev, dtype=torch.int16)
    w = torch.zeros(256, 9, 64, device=dev)
    with pytest.raises(L.WgsError):        # too few tiles for the 256 x 256 form (and not a multiple of 256 rows... the library declines loudly)
        C.conv2d(x, w, 3, pad=1, out=torch.empty(1, 16, 16, 256, device=dev), precision=2, w_split=C.split_weight(w, 2), a_amax=torch.ones(1, device=dev), x_f16=True,
                 rgb=dict(out=torch.empty(1, 16, 16, 4, device=dev), s=torch.ones(1, 256, device=dev), ld=256, w=torch.zeros(3, 256, device=dev), scale=1.0))
    w5 = torch.zeros(512, 9, 64, device=dev)
    with pytest.raises(L.WgsError):        # 512 output channels: no tile holds them all
        C.conv2d(x, w5, 3, pad=1, out=torch.empty(1, 16, 16, 512, device=dev), precision=2, w_split=C.split_weight(w5, 2), a_amax=torch.ones(1, device=dev), x_f16=True,
                 rgb=dict(out=torch.empty(1, 16, 16, 4, device=dev), s=torch.ones(1, 512, device=dev), ld=512, w=torch.zeros(3, 512, device=dev), scale=1.0))


def test_generator_same_image_with_and_without_the_fused_torgb(dev, monkeypatch):
    from warpedganspace_amd.gan_load import build_stylegan2
    torch.manual_seed(0)
    G = build_stylegan2(None, resolution=256).to(dev).eval()
    z = torch.randn(32, 512, device=dev)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(C, 'RGB_FUSED', on)
        with torch.no_grad():
            a = G(z, precision='mixed').clone()
        zz = z.clone().requires_grad_(True)
        img = G(zz, precision='mixed')
        img.backward(torch.linspace(-1, 1, img.numel(), device=dev).view_as(img))
        res[on] = (a, img.detach().clone(), zz.grad.clone())
    assert torch.equal(res[True][0], res[True][1])             # the pass that stores nothing and the pass that saves: same image
    e = float((res[True][1] - res[False][1]).abs().max() / res[False][1].abs().max())
    g = float((res[True][2] - res[False][2]).abs().max() / res[False][2].abs().max())
    print('fused ToRGB on vs off: image %.2e, gradient %.2e' % (e, g))
    assert e <= 3e-6 and g <= 1e-4
