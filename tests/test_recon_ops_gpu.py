"""GPU: Reconstructor glue kernels one by one at BASELINE-like row counts, against torch CPU float64
(the ReLU gate / pooling winner is taken from the HIP forward so that both sides differentiate the same
piecewise-linear function)."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_err
from warpedganspace_amd import _lib as L
from warpedganspace_amd.reconstructor import _BN

pytestmark = pytest.mark.gpu
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
nchw = lambda t: t.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize('B,Cc,H', [(4, 64, 16), (8, 64, 32), (3, 128, 12), (16, 64, 64), (2, 512, 3), (5, 84, 1)])
def test_bn_residual_relu_fwd_bwd(dev, B, Cc, H):
    torch.manual_seed(B + Cc)
    x = torch.randn(B, Cc, H, H, dtype=torch.float64, requires_grad=True)
    ga = (torch.randn(Cc, dtype=torch.float64) * 0.2 + 1).requires_grad_(True)
    be = torch.randn(Cc, dtype=torch.float64, requires_grad=True)
    res = torch.randn(B, Cc, H, H, dtype=torch.float64, requires_grad=True)
    rm, rv = torch.zeros(Cc, dtype=torch.float64), torch.ones(Cc, dtype=torch.float64)
    pre = F.batch_norm(x, rm, rv, ga, be, training=True, momentum=0.1, eps=1e-5) + res
    bn = torch.nn.BatchNorm2d(Cc).to(dev)
    bn.weight.data, bn.bias.data = ga.detach().float().to(dev), be.detach().float().to(dev)
    ws = torch.zeros(64 * 512, dtype=torch.float64, device=dev)
    xd, rd = nhwc(x.detach().float()).to(dev), nhwc(res.detach().float()).to(dev)
    yd, st = _BN.fwd(bn, xd, ws, residual=rd, relu=True, train=True)
    assert rel_err(nchw(yd), F.relu(pre).detach()) < 1e-6
    assert rel_err(bn.running_mean, rm) < 1e-5 and rel_err(bn.running_var, rv) < 1e-5 and int(bn.num_batches_tracked) == 1
    gate = nchw(yd > 0).cpu().double()                      # shared gate
    g1, g2 = torch.randn_like(pre), torch.randn_like(pre)
    (pre * gate * (g1 + g2)).sum().backward()
    dx, dres, dg, db = _BN.bwd(bn, xd, st, nhwc(g1.float()).to(dev), nhwc(g2.float()).to(dev), yd, ws, want_res=True)
    assert rel_err(nchw(dx), x.grad) < 2e-5
    assert rel_err(nchw(dres), res.grad) < 1e-6
    assert rel_err(dg, ga.grad) < 2e-5 and rel_err(db, be.grad) < 2e-5


@pytest.mark.parametrize('B,Cc,H,k,s,p', [(4, 64, 32, 3, 2, 1), (8, 64, 64, 3, 2, 1), (2, 8, 28, 2, 2, 0), (3, 16, 10, 2, 2, 0)])
def test_maxpool_fwd_bwd(dev, B, Cc, H, k, s, p):
    torch.manual_seed(H)
    x = F.relu(torch.randn(B, Cc, H, H)).requires_grad_(True)     # many exact ties at 0, like after ReLU
    y = F.max_pool2d(x, k, s, p)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    xd = nhwc(x.detach()).to(dev)
    Ho = y.shape[2]
    yd = torch.empty(B, Ho, Ho, Cc, device=dev)
    idx = torch.empty(B, Ho, Ho, Cc, dtype=torch.uint8, device=dev)
    L.check(L.lib().wgs_maxpool_fwd(L.ptr(xd), L.ptr(yd), L.rawptr(idx), B, H, H, Cc, k, s, p, L.stream()))
    dx = torch.empty_like(xd)
    gd = nhwc(g).to(dev)
    L.check(L.lib().wgs_maxpool_bwd(L.ptr(gd), L.rawptr(idx), L.ptr(dx), B, H, H, Cc, k, s, p, L.stream()))
    assert torch.equal(nchw(yd).cpu(), y.detach())
    assert rel_err(nchw(dx), x.grad) < 1e-6          # includes torch's first-maximum tie-breaking


def test_avgpool_pack_colsum(dev):
    B, P, Cc = 5, 64, 512
    x = torch.randn(B, P, Cc)
    xd = x.to(dev)
    y = torch.empty(B, Cc, device=dev)
    L.check(L.lib().wgs_avgpool_fwd(L.ptr(xd), L.ptr(y), B, P, Cc, L.stream()))
    assert rel_err(y, x.mean(1)) < 1e-6
    dx = torch.empty(B, P, Cc, device=dev)
    L.check(L.lib().wgs_avgpool_bwd(L.ptr(y), L.ptr(dx), B, P, Cc, L.stream()))
    assert rel_err(dx, (y.cpu() / P)[:, None, :].expand(B, P, Cc)) < 1e-6
    a, b = torch.randn(3, 3, 7, 9), torch.randn(3, 3, 7, 9)
    out = torch.empty(3, 7, 9, 8, device=dev)
    ad, bd = a.to(dev), b.to(dev)
    L.check(L.lib().wgs_pack_pair_nhwc(L.ptr(ad), L.ptr(bd), L.ptr(out), 3, 3, 63, 8, L.stream()))
    ref = torch.cat([a, b, torch.zeros(3, 2, 7, 9)], 1).permute(0, 2, 3, 1)
    assert torch.equal(out.cpu(), ref)
    d1, d2 = torch.empty(3, 3, 7, 9, device=dev), torch.empty(3, 3, 7, 9, device=dev)
    L.check(L.lib().wgs_unpack_pair_grad(L.ptr(out), L.ptr(d1), L.ptr(d2), 3, 3, 63, 8, L.stream()))
    assert torch.equal(d1.cpu(), a) and torch.equal(d2.cpu(), b)
    xs = torch.randn(1000, 64)
    cs = torch.empty(64, device=dev)
    ws = torch.zeros(64 * 64, dtype=torch.float64, device=dev)
    xsd = xs.to(dev)
    L.check(L.lib().wgs_colsum(L.ptr(xsd), L.ptr(cs), L.rawptr(ws), L.c_int64(1000), 64, L.stream()))
    assert rel_err(cs, xs.double().sum(0)) < 1e-6
