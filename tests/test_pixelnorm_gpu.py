"""GPU: PixelNorm forward / backward kernels (models/ProgGAN/model.py PixelNormLayer, models/StyleGAN2/model.py:9-15) against float64
torch, for the short channel rows of ProgGAN's feature maps (lane-group form) and the long latent rows (wave-per-row form)."""
import pytest
import torch

from tests.util import rel_err
from warpedganspace_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('rows,d', [(1000, 16), (777, 32), (4099, 64), (130, 128), (65, 256), (9, 512), (33, 24), (3, 2048)])
def test_pixelnorm_fwd_bwd(dev, rows, d):
    torch.manual_seed(rows + d)
    x = (torch.randn(rows, d) * 3).requires_grad_(True)
    g = torch.randn(rows, d)
    xd = x.detach().double().requires_grad_(True)
    y_ref = xd * torch.rsqrt(xd.pow(2).mean(1, keepdim=True) + 1e-8)
    y_ref.backward(g.double())
    lib, st = L.lib(), L.stream()
    xg, gg = x.detach().to(dev), g.to(dev)
    y, gx = torch.full_like(xg, 9.0), torch.full_like(xg, 9.0)
    L.check(lib.wgs_pixelnorm_fwd(L.ptr(xg), L.ptr(y), rows, d, L.c_float(1e-8), st), 'pn_fwd')
    L.check(lib.wgs_pixelnorm_bwd(L.ptr(xg), L.ptr(gg), L.ptr(gx), rows, d, L.c_float(1e-8), st), 'pn_bwd')
    assert rel_err(y, y_ref.detach()) < 1e-6 and rel_err(gx, xd.grad) < 2e-6


def test_pixelnorm_bwd_with_folded_leaky_relu_gate(dev):
    """wgs_pixelnorm_bwd_act = PixelNorm backward times the leaky-relu gate of its input (the previous block's activation backward)."""
    torch.manual_seed(5)
    rows, d = 515, 32
    x, g = torch.randn(rows, d), torch.randn(rows, d)
    xd = x.double().requires_grad_(True)
    (xd * torch.rsqrt(xd.pow(2).mean(1, keepdim=True) + 1e-8)).backward(g.double())
    ref = xd.grad * torch.where(x > 0, 1.0, 0.2).double()
    lib, st = L.lib(), L.stream()
    xg, gg, gx = x.to(dev), g.to(dev), torch.empty(rows, d, device=dev)
    L.check(lib.wgs_pixelnorm_bwd_act(L.ptr(xg), L.ptr(gg), L.ptr(gx), rows, d, L.c_float(1e-8), L.c_float(0.2), st), 'pn_bwd_act')
    assert rel_err(gx, ref) < 2e-6, rel_err(gx, ref)
