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
    with C.resolved(C.precision_code('f16')):
        for on in (True, False):
            monkeypatch.setattr(C, 'BLUR_BWD_F16', on)
            zz = z.clone().requires_grad_(True)
            img = G([zz])[0]
            gi = torch.linspace(-1, 1, img.numel(), device=dev).view_as(img)
            img.backward(gi)
            outs.append(zz.grad.clone())
    assert torch.isfinite(outs[0]).all()
    # (the style-gradient reductions behind z use fp32 atomics: two runs of the SAME route differ in the last bits)
    assert (outs[0] - outs[1]).abs().max() <= 1e-5 * outs[1].abs().max()
