"""GPU: ToRGB with the skip branch's Upsample evaluated in its epilogue (wgs_sg2_torgb_up_fwd) against the two-launch form
(wgs_upfirdn2d + wgs_sg2_torgb_fwd) and a float64 statement of models/StyleGAN2/model.py:257-282."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_err
from warpedganspace_amd import _lib as L
from warpedganspace_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,H,C', [(2, 8, 32), (3, 16, 64), (1, 32, 512), (2, 64, 128)])
def test_torgb_with_inplace_skip_upsample(dev, B, H, C):
    torch.manual_seed(H + C)
    x = torch.randn(B, H, H, C)
    s = torch.randn(B, C) + 1.0
    w = torch.randn(3, C) / C ** 0.5
    bias = torch.randn(3) * 0.1
    skip = torch.randn(B, 3, H // 2, H // 2)
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    upk = k1[:, None] * k1[None, :] / 64.0 * 4.0
    wscale = 1.0 / C ** 0.5
    # float64: 1x1 modulated conv (no demodulation) + bias + upfirdn2d(skip, up=2, pad=(2,1))
    rgb = torch.einsum('bhwc,bc,oc->bohw', x.double(), s.double(), w.double()) * wscale + bias.double()[None, :, None, None]
    z = torch.zeros(B, 3, H, H, dtype=torch.float64)
    z[:, :, ::2, ::2] = skip.double()
    zp = F.pad(z, (2, 1, 2, 1))
    kf = torch.flip(upk.double(), [0, 1])[None, None].repeat(3, 1, 1, 1)
    ref = rgb + F.conv2d(zp, kf, groups=3)
    xd, sd, wd, bd, kd, skd = (t.to(dev).contiguous() for t in (x, s, w, bias, upk, skip))
    lib, st = L.lib(), L.stream()
    img = torch.empty(B, 3, H, H, device=dev)
    L.check(lib.wgs_sg2_torgb_up_fwd(L.ptr(xd), L.ptr(sd), 0, L.ptr(wd), L.ptr(bd), L.ptr(skd), L.ptr(kd), L.ptr(img), B, H, H, C,
                                     L.c_float(wscale), st), 'torgb_up')
    assert rel_err(img, ref) < 2e-6
    up = ops.upfirdn2d_mhwc(skd.reshape(B * 3, H // 2, H // 2, 1), kd, 2, 2, 1, 1, 2, 1, 2, 1).reshape(B, 3, H, H)
    img2 = torch.empty_like(img)
    L.check(lib.wgs_sg2_torgb_fwd(L.ptr(xd), L.ptr(sd), L.ptr(wd), L.ptr(bd), L.ptr(up), L.ptr(img2), B, H * H, C,
                                  L.c_float(wscale), st), 'torgb')
    assert rel_err(img, img2.cpu()) < 1e-6
    # the style vector as rows of a wider matrix (the generator's [B, sumC] modulation output): s_ld
    wide = torch.randn(B, C + 24, device=dev)
    wide[:, 8:8 + C] = sd
    img3 = torch.empty_like(img)
    L.check(lib.wgs_sg2_torgb_up_fwd(L.ptr(xd), L.rawptr(wide[:, 8:]), C + 24, L.ptr(wd), L.ptr(bd), L.ptr(skd), L.ptr(kd), L.ptr(img3),
                                     B, H, H, C, L.c_float(wscale), st), 'torgb_up strided')
    assert torch.equal(img3, img)
