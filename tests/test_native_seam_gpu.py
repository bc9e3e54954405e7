"""GPU: the two reference native ops (fused_bias_act, upfirdn2d) re-done in HIP, through the C ABI."""
import pytest
import torch

from oracle import wgs_oracle as O
from tests import golden_inputs as GI
from tests.util import rel_err
from warpedganspace_amd import ops

pytestmark = pytest.mark.gpu


def test_upfirdn2d_vs_reference_native_golden(dev, golden):
    g = golden('native_ops')
    k = O.make_blur_kernel()
    for i, c in enumerate(GI.UPFIRDN_CASES):
        x = GI.rt(100 + i, c['major'], c['h'], c['w'], c['minor'])
        kk = k * c['gain'] if c['name'] != 'up3_down2' else GI.rt(777, 5, 3)
        y = ops.upfirdn2d_mhwc(x.to(dev), kk.to(dev), c['up'], c['up'], c['down'], c['down'], *c['pad'])
        assert rel_err(y, g['upfirdn_' + c['name']]) < 1e-6, c['name']


@pytest.mark.parametrize('shape', [(4, 16, 9, 9), (2, 512), (3, 5, 7, 3), (1, 8, 64, 64)])
def test_bias_act_all_modes(dev, shape):
    x = torch.randn(*shape)
    b = torch.randn(shape[1])
    ref = torch.randn(*shape)
    for act, grad in ((3, 0), (3, 1), (1, 0), (1, 1), (3, 2)):
        for use_b in (True, False):
            if grad == 1 and use_b:
                continue
            y = ops.fused_bias_act(x.to(dev), b.to(dev) if use_b else None, ref.to(dev) if grad == 1 else None,
                                   act, grad, 0.2, 2 ** 0.5)
            yo = O.fused_bias_act(x, b if use_b else None, ref, act, grad, 0.2, 2 ** 0.5)
            assert rel_err(y, yo) < 1e-6, (act, grad, use_b)


def test_fused_leaky_relu_autograd(dev):
    x = torch.randn(4, 12, 6, 6, requires_grad=True)
    b = torch.randn(12, requires_grad=True)
    p = torch.randn(4, 12, 6, 6)
    O.fused_leaky_relu(x, b).mul(p).sum().backward()
    xd = x.detach().to(dev).requires_grad_(True)
    bd = b.detach().to(dev).requires_grad_(True)
    ops.fused_leaky_relu(xd, bd).mul(p.to(dev)).sum().backward()
    assert rel_err(xd.grad, x.grad) < 1e-6
    assert rel_err(bd.grad, b.grad) < 1e-5


def test_upfirdn2d_autograd_nchw(dev):
    k = O.make_blur_kernel() * 4
    for up, down, pad in ((1, 1, (1, 1)), (2, 1, (2, 1)), (1, 2, (1, 1))):
        x = torch.randn(2, 3, 8, 8, requires_grad=True)
        y = O.upfirdn2d(x, k, up=up, down=down, pad=pad)
        p = torch.randn_like(y)
        (y * p).sum().backward()
        xd = x.detach().to(dev).requires_grad_(True)
        yd = ops.upfirdn2d(xd, k.to(dev), up=up, down=down, pad=pad)
        (yd * p.to(dev)).sum().backward()
        assert rel_err(yd, y) < 1e-6
        assert rel_err(xd.grad, x.grad) < 1e-6


def test_bn_scratch_contract_is_checkable(dev):
    """wgs_bn_fwd / wgs_bn_bwd / wgs_colsum rely on their fp64 scratch being zero on entry (include/wgs.h); a dirty buffer gives
    wrong statistics silently — unless WGS_CHECK_WS=1, which makes every call verify it first and fail with WGS_EINVAL."""
    import os
    from warpedganspace_amd import _lib as L
    lib = L.lib()
    N, C = 512, 8
    x = torch.randn(N, C, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    y, mean, invstd = torch.empty_like(x), torch.empty(C, device=dev), torch.empty(C, device=dev)

    def bn(ws):
        return lib.wgs_bn_fwd(L.ptr(x), L.ptr(g), L.ptr(b), None, L.ptr(y), L.ptr(mean), L.ptr(invstd), None, None, None, L.rawptr(ws),
                              L.c_int64(N), C, L.c_float(1e-5), L.c_float(0.1), 0, 1, L.stream())
    ws = torch.zeros(64 * C, dtype=torch.float64, device=dev)
    assert bn(ws) == 0 and bn(ws) == 0                     # left zero on exit: the second call needs no memset
    torch.cuda.synchronize()
    assert float(ws.abs().max()) == 0.0 and float((mean - x.mean(0)).abs().max()) < 1e-6
    os.environ['WGS_CHECK_WS'] = '1'
    lib.wgs_dev_reload_flags()
    try:
        assert bn(ws) == 0
        dirty = torch.zeros(64 * C, dtype=torch.float64, device=dev)
        dirty[5] = 1.0
        assert bn(dirty) == -22 and b'not zero on entry' in lib.wgs_last_error()
        out = torch.empty(C, device=dev)
        assert lib.wgs_colsum(L.ptr(x), L.ptr(out), L.rawptr(dirty), L.c_int64(N), C, L.stream()) == -22
    finally:
        del os.environ['WGS_CHECK_WS']
        lib.wgs_dev_reload_flags()
