"""GPU: the fused, batched self-attention core (csrc/attention.hip: wgs_attn_fwd / wgs_attn_bwd) against a float64 statement of
models/BigGAN/layers.py:157-166 (torch.bmm + F.softmax + torch.bmm and their autograd backward), at BigGAN-128's and BigGAN-256's
channel counts."""
import pytest
import torch

from tests.util import rel_err
from warpedganspace_amd import _lib as L

pytestmark = pytest.mark.gpu


def reference(theta, phi, g, d_o):
    th, ph, gg = (t.double().clone().requires_grad_(True) for t in (theta, phi, g))
    beta = torch.softmax(torch.bmm(th, ph.transpose(1, 2)), -1)          # [B, Pq, Pk]
    o = torch.bmm(beta, gg)
    o.backward(d_o.double())
    return o.detach(), th.grad, ph.grad, gg.grad, torch.logsumexp(torch.bmm(th, ph.transpose(1, 2)), -1).detach()


@pytest.mark.parametrize('B,H,c8,c2,scale', [(2, 32, 24, 96, 1.0), (3, 64, 24, 96, 3.0), (2, 64, 96, 384, 1.0), (1, 32, 48, 192, 0.2)])
def test_attention_fwd_bwd_vs_fp64(dev, B, H, c8, c2, scale):
    torch.manual_seed(H + c8)
    Pq, Pk = H * H, H * H // 4
    theta = torch.randn(B, Pq, c8) * scale / c8 ** 0.5            # scale 3: peaked rows (max-subtraction matters)
    phi, g, d_o = torch.randn(B, Pk, c8), torch.randn(B, Pk, c2), torch.randn(B, Pq, c2)
    o_ref, dth_ref, dph_ref, dg_ref, lse_ref = reference(theta, phi, g, d_o)
    lib = L.lib()
    assert lib.wgs_attn_supported(B, Pq, Pk, c8, c2) == 1
    td, pd, gd, dod = theta.to(dev), phi.to(dev), g.to(dev), d_o.to(dev)
    o, lse = torch.empty(B, Pq, c2, device=dev), torch.empty(B, Pq, device=dev)
    L.check(lib.wgs_attn_fwd(L.ptr(td), L.ptr(pd), L.ptr(gd), L.ptr(o), L.ptr(lse), B, Pq, Pk, c8, c2, L.stream()), 'attn_fwd')
    assert rel_err(o, o_ref) < 1e-5 and rel_err(lse, lse_ref) < 1e-5
    dth, dph, dg = torch.full_like(td, 7.0), torch.full_like(pd, 7.0), torch.full_like(gd, 7.0)      # overwritten, not accumulated
    ws = torch.empty(B * Pq, device=dev)
    L.check(lib.wgs_attn_bwd(L.ptr(td), L.ptr(pd), L.ptr(gd), L.ptr(o), L.ptr(lse), L.ptr(dod), L.ptr(ws), L.ptr(dth), L.ptr(dph),
                             L.ptr(dg), B, Pq, Pk, c8, c2, L.stream()), 'attn_bwd')
    print('attention B=%d %dx%d c8=%d c2=%d: o %.1e  dtheta %.1e  dphi %.1e  dg %.1e' % (
        B, H, H, c8, c2, rel_err(o, o_ref), rel_err(dth, dth_ref), rel_err(dph, dph_ref), rel_err(dg, dg_ref)))
    assert rel_err(dth, dth_ref) < 2e-5 and rel_err(dph, dph_ref) < 2e-5 and rel_err(dg, dg_ref) < 2e-5
    # deterministic: no atomics anywhere
    dth2, dph2, dg2 = torch.empty_like(td), torch.empty_like(pd), torch.empty_like(gd)
    L.check(lib.wgs_attn_bwd(L.ptr(td), L.ptr(pd), L.ptr(gd), L.ptr(o), L.ptr(lse), L.ptr(dod), L.ptr(ws), L.ptr(dth2), L.ptr(dph2),
                             L.ptr(dg2), B, Pq, Pk, c8, c2, L.stream()), 'attn_bwd')
    assert torch.equal(dth, dth2) and torch.equal(dph, dph2) and torch.equal(dg, dg2)


def test_unsupported_shapes_are_refused(dev):
    lib = L.lib()
    assert lib.wgs_attn_supported(2, 4096, 1024, 32, 128) == 0 and lib.wgs_attn_supported(2, 100, 25, 24, 96) == 0
    t = torch.zeros(1, 128, 32, device=dev)
    rc = lib.wgs_attn_fwd(L.ptr(t), L.ptr(t), L.ptr(t), L.ptr(t), L.ptr(t), 1, 128, 64, 32, 128, L.stream())
    assert rc != 0 and b'unsupported shape' in lib.wgs_last_error()
