"""GPU: the StyleGAN2 mapping network in one launch (wgs_mapping_mlp_fwd) — bit-identical to PixelNorm + 8 x EqualLinear launches,
and within fp32 rounding of a float64 statement of models/StyleGAN2/model.py:288-295."""
import ctypes

import pytest
import torch

from tests.util import rel_err
from warpedganspace_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,nl', [(32, 8), (5, 8), (64, 3), (1, 1)])
def test_mapping_mlp_one_launch(dev, B, nl):
    torch.manual_seed(B + nl)
    d, lr_mul = 512, 0.01
    scale = (1.0 / d ** 0.5) * lr_mul
    z = torch.randn(B, d, device=dev)
    ws = [torch.randn(d, d, device=dev) / lr_mul for _ in range(nl)]
    bs = [torch.randn(d, device=dev) for _ in range(nl)]
    lib, st = L.lib(), L.stream()
    # per-layer launches
    x = torch.empty_like(z)
    L.check(lib.wgs_pixelnorm_fwd(L.ptr(z), L.ptr(x), B, d, L.c_float(1e-8), st), 'pn')
    ref = [x]
    for w, b in zip(ws, bs):
        y = torch.empty(B, d, device=dev)
        L.check(lib.wgs_linear_fwd(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, d, d, d, d, L.c_float(scale), L.c_float(lr_mul), 0, 1,
                                   L.c_float(0.0), L.c_float(1.0), st), 'lin')
        ref.append(y)
        x = y
    # one launch
    acts = torch.full((nl + 1, B, d), float('nan'), device=dev)
    wp = (ctypes.c_void_p * nl)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * nl)(*[b.data_ptr() for b in bs])
    L.check(lib.wgs_mapping_mlp_fwd(L.ptr(z), wp, bp, L.ptr(acts), B, d, nl, L.c_float(scale), L.c_float(lr_mul), L.c_float(1e-8), st),
            'mlp')
    for l in range(nl + 1):
        assert torch.equal(acts[l], ref[l]), l
    # float64 statement
    x64 = z.double().cpu()
    x64 = x64 * torch.rsqrt((x64 ** 2).mean(1, keepdim=True) + 1e-8)
    for w, b in zip(ws, bs):
        x64 = torch.nn.functional.leaky_relu(x64 @ (w.double().cpu() * scale).t() + b.double().cpu() * lr_mul, 0.2) * 2 ** 0.5
    assert rel_err(acts[nl], x64) < 1e-5


def test_mapping_mlp_rejects_other_widths(dev):
    z = torch.zeros(2, 256, device=dev)
    w = torch.zeros(256, 256, device=dev)
    b = torch.zeros(256, device=dev)
    acts = torch.zeros(2, 2, 256, device=dev)
    with pytest.raises(L.WgsError):
        L.check(L.lib().wgs_mapping_mlp_fwd(L.ptr(z), (ctypes.c_void_p * 1)(w.data_ptr()), (ctypes.c_void_p * 1)(b.data_ptr()),
                                            L.ptr(acts), 2, 256, 1, L.c_float(1.0), L.c_float(1.0), L.c_float(1e-8), L.stream()), 'mlp')


@pytest.mark.parametrize('B,nl', [(32, 8), (5, 8), (3, 2)])
def test_mapping_mlp_backward_one_launch(dev, B, nl):
    """wgs_mapping_mlp_bwd == nl x wgs_linear_dgrad through the fused leaky-relu gates (models/StyleGAN2/model.py:127-129 backward),
    against the per-layer launches (fp32 summation order) and a float64 autograd statement."""
    torch.manual_seed(B * 7 + nl)
    d, lr_mul = 512, 0.01
    scale = (1.0 / d ** 0.5) * lr_mul
    z = torch.randn(B, d, device=dev)
    ws = [torch.randn(d, d, device=dev) / lr_mul for _ in range(nl)]
    bs = [torch.randn(d, device=dev) for _ in range(nl)]
    lib, st = L.lib(), L.stream()
    acts = torch.empty(nl + 1, B, d, device=dev)
    wp = (ctypes.c_void_p * nl)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * nl)(*[b.data_ptr() for b in bs])
    L.check(lib.wgs_mapping_mlp_fwd(L.ptr(z), wp, bp, L.ptr(acts), B, d, nl, L.c_float(scale), L.c_float(lr_mul), L.c_float(1e-8), st), 'mlp')
    gw = torch.randn(B, d, device=dev)
    g = gw
    for i in range(nl - 1, -1, -1):
        gx = torch.empty(B, d, device=dev)
        L.check(lib.wgs_linear_dgrad(L.ptr(g), L.ptr(ws[i]), L.ptr(acts[i + 1]), L.ptr(gx), B, d, d, d, d, L.c_float(scale), L.c_float(0.2),
                                     L.c_float(2 ** 0.5), 0, st), 'dgrad')
        g = gx
    one = torch.full((B, d), float('nan'), device=dev)
    L.check(lib.wgs_mapping_mlp_bwd(L.ptr(gw), wp, L.ptr(acts), L.ptr(one), B, d, nl, L.c_float(scale), st), 'mlp_bwd')
    assert rel_err(one, g) < 2e-6
    x0 = acts[0].double().cpu().requires_grad_(True)
    x64 = x0
    for w, b in zip(ws, bs):
        x64 = torch.nn.functional.leaky_relu(x64 @ (w.double().cpu() * scale).t() + b.double().cpu() * lr_mul, 0.2) * 2 ** 0.5
    x64.backward(gw.double().cpu())
    assert rel_err(one, x0.grad) < 1e-5


def test_generator_gradient_same_with_fused_mapping_backward(dev, monkeypatch):
    from tests import golden_inputs as GI
    from warpedganspace_amd import stylegan2 as SG
    torch.manual_seed(0)
    G = SG.Generator(32, 512, 8)
    G.load_state_dict(GI.fill_state_dict(G.state_dict(), 977))
    G = G.to(dev).eval()
    z = torch.randn(6, 512, device=dev)
    out = []
    for on in (True, False):
        monkeypatch.setattr(SG, 'MAPPING_BWD_FUSED', on)
        zz = z.clone().requires_grad_(True)
        img = G([zz])[0]
        img.backward(torch.linspace(-1, 1, img.numel(), device=dev).view_as(img))
        out.append(zz.grad.clone())
    assert rel_err(out[0], out[1]) < 1e-5
