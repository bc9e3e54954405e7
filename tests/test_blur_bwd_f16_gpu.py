"""GPU: the transposed blur of an up-sampling layer stored as the fp16 operand plane of its gradient conv
(wgs_sg2_blur_bwd_f16 -> wgs_conv_desc.x_f16; backward of models/StyleGAN2/model.py:201-212 + Blur, op/upfirdn2d.py:110-115).

  * the plane holds exactly f16_rn(dt * 2^k) of the fp32 route's dt = upfirdn2d(dy, flip(k), pad (2,2)), k from (a_amax, a_bound);
  * the gradient conv fed with the plane (LDS-DMA kernel, no pre-pass) returns the SAME BITS as the conv fed with the fp32 dt
    wherever that one also runs 256-row tiles over the whole K (the shapes the generator uses the route for); small launches,
    whose fp32 route splits K, agree to fp32 summation order;
  * a generator backward with the route on and off returns the same gradient (to the order of its atomic reductions)."""
import pytest
import torch

from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C
from warpedganspace_amd import ops

pytestmark = pytest.mark.gpu


def blur_f():
    k = torch.tensor([1.0, 3.0, 3.0, 1.0])
    k2 = k[:, None] * k[None, :]
    return (k2 / k2.sum() * 4.0).flip(0, 1).contiguous()


@pytest.mark.parametrize('B,Cdy,Cdx,H,mag,same_bits', [(32, 64, 256, 64, 1.0, True), (16, 128, 512, 64, 3e-6, True), (4, 128, 256, 32, 1.0, False),
                                                      (1, 64, 128, 40, 7e4, False), (3, 32, 128, 10, 1.0, False)])
def test_plane_bits_and_conv_bits(dev, B, Cdy, Cdx, H, mag, same_bits):
    torch.manual_seed(B * 100 + H)
    dy = (torch.randn(B, H, H, Cdy, device=dev) * mag).contiguous()
    kf = blur_f().to(dev)
    am = dy.abs().max().reshape(1) * 1.3            # any over-estimate of max|dy|
    dt = ops.upfirdn2d_mhwc(dy, kf, 1, 1, 1, 1, 2, 2, 2, 2)
    plane = C.blur_bwd_f16(dy, kf, am, 4.0)
    assert plane.shape == dt.shape and plane.dtype == torch.int16
    # the scale of conv_scheme.h: amax * bound * 2^k in [2^11, 2^12)
    a = float(am.item()) * 4.0
    k = 0
    while a * 2.0 ** k < 2048.0:
        k += 1
    while a * 2.0 ** k >= 4096.0:
        k -= 1
    want = (dt * 2.0 ** k).half().view(torch.int16)
    assert torch.equal(plane, want)
    # gradient conv: plane route vs fp32 route (register-staged or pre-pass + DMA, whatever the library picks)
    wp = C.pack_weight(torch.randn(Cdy, Cdx, 3, 3) / (9 * Cdx) ** 0.5).to(dev)     # forward weight [Co=Cdy, Ci=Cdx, 3, 3] packed
    wt = C.repack_w_t(wp, Cdy, 9, Cdx)
    wts = C.split_weight(wt, 2)
    g_ref = C.conv_transpose2d_s2_dgrad(dt, wt, w_split=wts, a_amax=am, a_bound=4.0, precision=2)
    g_new = C.conv_transpose2d_s2_dgrad(plane, wt, w_split=wts, a_amax=am, a_bound=4.0, precision=2, x_f16=True)
    assert g_new.shape == (B, (H + 1 - 3) // 2 + 1, (H + 1 - 3) // 2 + 1, Cdx)
    if same_bits:
        assert torch.equal(g_new, g_ref)
    assert (g_new - g_ref).abs().max() <= 2e-6 * g_ref.abs().max()
    g32 = C.conv_transpose2d_s2_dgrad(dt, wt, precision=0)
    assert (g_new - g32).abs().max() <= 2e-3 * g32.abs().max()


def test_x_f16_argument_checks(dev):
    plane = torch.zeros(1, 9, 9, 32, device=dev, dtype=torch.int16)
    wt = torch.zeros(9, 128, 32, device=dev)
    am = torch.ones(1, device=dev)
    with pytest.raises(L.WgsError):          # needs precision 2 and pre-split weights
        C.conv_transpose2d_s2_dgrad(plane, wt, a_amax=am, a_bound=4.0, precision=3, x_f16=True)
    with pytest.raises(L.WgsError):
        C.conv_transpose2d_s2_dgrad(plane, wt, a_amax=am, a_bound=4.0, precision=2, x_f16=True)


def test_generator_backward_same_gradient(dev, monkeypatch):
    from tests import golden_inputs as GI
    from warpedganspace_amd.stylegan2 import Generator
    torch.manual_seed(0)
    G = Generator(128, 512, 8)
    G.load_state_dict(GI.fill_state_dict(G.state_dict(), 977))
    G = G.to(dev).eval()
    z = torch.randn(32, 512, device=dev)
    outs = []
    monkeypatch.setattr(C, 'DY_PLANE_UP', False)      # (the route of the test below is not bit-identical to the fp32 route)
    for on in (True, False):
        monkeypatch.setattr(C, 'BLUR_BWD_F16', on)
        zz = z.clone().requires_grad_(True)
        img = G([zz], precision='f16')[0]
        gi = torch.linspace(-1, 1, img.numel(), device=dev).view_as(img)
        img.backward(gi)
        outs.append(zz.grad.clone())
    assert torch.isfinite(outs[0]).all()
    # (the style-gradient reductions behind z use fp32 atomics: two runs of the SAME route differ in the last bits)
    assert (outs[0] - outs[1]).abs().max() <= 1e-5 * outs[1].abs().max()


@pytest.mark.parametrize('B,C_,H,mag', [(8, 64, 64, 1.0), (4, 128, 32, 3e-6), (2, 32, 40, 7e4)])
def test_plane_from_a_plane_has_the_bits_of_the_fp32_route(dev, B, C_, H, mag):
    """wgs_sg2_blur_bwd_f16_x16: dy handed over as the fp16 plane f16_rn(dy * 2^k1) (k1 from a_amax alone) gives bit for bit the plane that
    wgs_sg2_blur_bwd_f16 makes of the fp32 tensor holding the same (fp16-representable) values."""
    torch.manual_seed(B * 10 + H)
    dy = (torch.randn(B, H, H, C_, device=dev) * mag)
    am = dy.abs().amax().reshape(1) * 1.7            # a bound, not the maximum
    k1 = 12 - (torch.floor(torch.log2(am)).item() + 1)      # am * 2^k1 in [2^11, 2^12)
    dyq = (dy * 2.0 ** k1).half()                            # the plane sg2_act_bwd_f16 would write
    dy_vals = (dyq.float() * 2.0 ** -k1).contiguous()        # the values it holds
    kf = blur_f().to(dev)
    want = C.blur_bwd_f16(dy_vals, kf, am, 4.0)
    got = C.blur_bwd_f16(dyq.view(torch.int16).contiguous(), kf, am, 4.0)
    assert got.dtype == torch.int16 and torch.equal(got, want)


def test_generator_backward_with_the_up_layers_dy_as_fp16_planes(dev, monkeypatch):
    """conv.DY_PLANE_UP: the activation backward of an up-sampling layer stores its dy as an fp16 plane for the transposed blur.  One more fp16
    rounding in front of a 16-tap average: the gradient moves by less than a fifth of the default arithmetic's own distance from the oracle
    (3.8e-4 / 6.0e-4)."""
    from tests import golden_inputs as GI
    from warpedganspace_amd.stylegan2 import Generator
    torch.manual_seed(0)
    G = Generator(128, 512, 8)
    G.load_state_dict(GI.fill_state_dict(G.state_dict(), 977))
    G = G.to(dev).eval()
    z = torch.randn(32, 512, device=dev)
    outs = []
    for on in (True, False):
        monkeypatch.setattr(C, 'DY_PLANE_UP', on)
        zz = z.clone().requires_grad_(True)
        img = G([zz], precision='f16')[0]
        gi = torch.linspace(-1, 1, img.numel(), device=dev).view_as(img)
        img.backward(gi)
        outs.append(zz.grad.clone())
    assert torch.isfinite(outs[0]).all()
    d = float((outs[0] - outs[1]).abs().max() / outs[1].abs().max())
    print('DY_PLANE_UP on vs off: %.2e' % d)
    assert 0.0 < d <= 1.2e-4


@pytest.mark.parametrize('with_gA,with_rgb', [(True, True), (True, False), (False, True)])
def test_act_bwd_plane_and_bound(dev, with_gA, with_rgb):
    """wgs_sg2_act_bwd_f16 stores f16_rn(dy * 2^k) of wgs_sg2_act_bwd's dy (k from the a-priori bound), the same reductions, and
    wgs_sg2_dy_bound is an upper bound of max|dy| that is not absurdly loose."""
    torch.manual_seed(7 + with_gA * 2 + with_rgb)
    lib = L.lib()
    B, H, Cc, sumC = 3, 24, 64, 200
    P = H * H
    out = torch.randn(B, P, Cc, device=dev)
    gA = torch.randn(B, P, Cc, device=dev) * 3e-4 if with_gA else None
    S = torch.randn(B, sumC, device=dev) + 1.0
    sA, sR = S[:, 10:], S[:, 100:]
    drgb = torch.randn(B, 3, P, device=dev) * 2e-3 if with_rgb else None
    wR = torch.randn(3, Cc, device=dev)
    noise, nw, bias = torch.randn(P, device=dev), torch.full((1,), 0.3, device=dev), torch.randn(Cc, device=dev) * 0.1
    demod = torch.rand(B, Cc, device=dev) + 0.5
    rscale = 0.125

    def run(f16, bound=None):
        num, dsA, dsR = (torch.zeros(B, Cc, device=dev) for _ in range(3))
        am = torch.zeros(1, device=dev)
        common = (L.ptr(out), L.ptr(gA), L.rawptr(sA if with_gA else None), L.ptr(drgb), L.ptr(wR) if with_rgb else None,
                  L.rawptr(sR if with_rgb else None), L.c_float(rscale if with_rgb else 0.0), L.ptr(noise), L.ptr(nw), L.ptr(bias))
        tail = (L.ptr(num), L.ptr(dsA) if with_gA else None, L.ptr(dsR) if with_rgb else None, L.ptr(demod), L.ptr(am), B, P, Cc, sumC, L.stream())
        if f16:
            dy = torch.empty(B, P, Cc, device=dev, dtype=torch.int16)
            L.check(lib.wgs_sg2_act_bwd_f16(*common, L.ptr(dy, torch.int16), L.ptr(bound), *tail), 'act_bwd_f16')
        else:
            dy = torch.empty(B, P, Cc, device=dev)
            L.check(lib.wgs_sg2_act_bwd(*common, L.ptr(dy), *tail), 'act_bwd')
        return dy, num, dsA, dsR, am

    dy, num, dsA, dsR, am = run(False)
    bound = torch.zeros(1, device=dev)
    gmax = gA.abs().amax().reshape(1) if with_gA else None
    dmax = (drgb.abs().amax().reshape(1) / 4.0) if with_rgb else None           # given as max/4 with factor 4
    L.check(lib.wgs_sg2_dy_bound(L.ptr(gmax), L.rawptr(sA if with_gA else None), L.ptr(dmax), L.c_float(4.0), L.ptr(wR) if with_rgb else None,
                                 L.rawptr(sR if with_rgb else None), L.c_float(rscale if with_rgb else 0.0), L.ptr(demod), L.ptr(bound),
                                 B, Cc, sumC, L.stream()), 'dy_bound')
    assert am.item() <= bound.item() <= 300.0 * am.item(), (am.item(), bound.item())
    plane, num2, dsA2, dsR2, am2 = run(True, bound)
    a, k = bound.item(), 0
    while a * 2.0 ** k < 2048.0:
        k += 1
    while a * 2.0 ** k >= 4096.0:
        k -= 1
    assert torch.equal(plane, (dy * 2.0 ** k).half().view(torch.int16))
    assert am2.item() == am.item()
    for u, v in ((num, num2), (dsA, dsA2), (dsR, dsR2)):
        assert (u - v).abs().max() <= 1e-5 * (u.abs().max() + 1e-30)


def test_generator_backward_dy_plane_same_gradient(dev, monkeypatch):
    from tests import golden_inputs as GI
    from warpedganspace_amd.stylegan2 import Generator
    torch.manual_seed(0)
    G = Generator(128, 512, 8)
    G.load_state_dict(GI.fill_state_dict(G.state_dict(), 977))
    G = G.to(dev).eval()
    z = torch.randn(32, 512, device=dev)
    outs = []
    for on in (True, False):
        monkeypatch.setattr(C, 'DY_PLANE', on)
        zz = z.clone().requires_grad_(True)
        img = G([zz], precision='f16')[0]
        gi = torch.linspace(-1, 1, img.numel(), device=dev).view_as(img)
        img.backward(gi)
        outs.append(zz.grad.clone())
    assert torch.isfinite(outs[0]).all()
    # same fp16 roundings unless a value falls below the (looser) scale's normal range: < 2^-19 of the tensor's maximum
    assert (outs[0] - outs[1]).abs().max() <= 2e-5 * outs[1].abs().max()
