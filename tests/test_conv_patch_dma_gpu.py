"""GPU: the all-DMA form of the 128 x 128 patch tile (csrc/conv_patch_dma.hip) — the stride-1 3x3 conv of a producer-written fp16 plane in
plain fp16 (StyleGAN2's 128 -> 128 layers at 256^2; the reference's op is models/StyleGAN2/model.py:187-228 ModulatedConv2d.forward, the
style product and the rounding to fp16 having been done by the producing kernel).

  * same bits as the register-staged patch kernel it replaces (WGS_PATCH_NODMA=1: same tile, same order of the MFMA sums), for one to
    four 32-channel chunks, maps 16 .. 256 wide (one tile column, image borders on every side of a tile), odd batch sizes, both tile shapes;
  * against the convolution in fp64 within the plain-fp16 scheme's tolerance;
  * 256 output columns per workgroup (Cout % 256 == 0, opt-in): same bits as the register-staged 256 x 256 patch tile, fp32-rounding close to the LDS-DMA kernel;
  * launches the library declines for this kernel (too few tiles) still run, through the other kernels."""
import os

import pytest
import torch

from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C

pytestmark = pytest.mark.gpu
SQRT2 = 2.0 ** 0.5


def _flags(**env):
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    L.lib().wgs_dev_reload_flags()


def _case(dev, B, Ci, Co, H, seed):
    torch.manual_seed(seed)
    x = torch.randn(B, H, H, Ci, device=dev)
    bound = (x.abs().amax() * 1.7).reshape(1).contiguous()
    k = 0
    while bound.item() * 2.0 ** k < 2048.0:
        k += 1
    while bound.item() * 2.0 ** k >= 4096.0:
        k -= 1
    plane = (x * 2.0 ** k).half().view(torch.int16).contiguous()
    w = torch.randn(Co, 9, Ci, device=dev) / (9 * Ci) ** 0.5
    demod = torch.rand(B, Co, device=dev) + 0.5
    noise, nw, bias = torch.randn(H * H, device=dev), torch.full((1,), 0.3, device=dev), torch.randn(Co, device=dev) * 0.2
    epi = dict(col_scale=demod, noise=noise, noise_w=nw, bias=bias, act_slope=0.2, gain=SQRT2, w_split=C.split_weight(w, 2), precision=2,
               a_amax=bound, a_bound=1.0, x_f16=True)
    return x, plane, w, demod, noise, bias, epi


def _run(plane, w, epi, B, H, Co, dev):
    lib = L.lib()
    lib.wgs_dev_trace_kernels(1)
    try:
        am = torch.zeros(1, device=dev)
        y = C.conv2d(plane, w, 3, pad=1, out=torch.empty(B, H, H, Co, device=dev), y_amax=am, **epi)
        return y, am, lib.wgs_dev_last_kernel().decode()
    finally:
        lib.wgs_dev_trace_kernels(0)


@pytest.mark.parametrize('bm', [128, 256])
@pytest.mark.parametrize('B,Ci,H', [(32, 128, 64), (52, 32, 32), (4, 96, 128), (13, 64, 64), (200, 128, 16), (2, 128, 256)])
def test_same_bits_as_the_register_staged_patch_kernel_and_close_to_fp64(dev, B, Ci, H, bm):
    Co = 128
    if bm == 256 and H < 32:
        pytest.skip('the 256-row tile is 8 x 32 pixels')
    x, plane, w, demod, noise, bias, epi = _case(dev, B, Ci, Co, H, 7 * B + Ci + H)
    try:
        _flags(WGS_PLANE_PATCH_MAX_CO=100000, WGS_PATCH_DMA_BM=bm)
        got, am, sym = _run(plane, w, epi, B, H, Co, dev)
        assert sym.startswith('patch_dma_kernel<%d, 128, false>' % bm), sym
        _flags(WGS_PATCH_NODMA=1)
        ref, am_ref, sym_ref = _run(plane, w, epi, B, H, Co, dev)
        assert sym_ref.startswith('igemm_patch_kernel<1, 128, 128'), sym_ref
    finally:
        _flags(WGS_PATCH_NODMA=None, WGS_PLANE_PATCH_MAX_CO=None, WGS_PATCH_DMA_BM=None)
    assert torch.equal(got, ref)
    assert am.item() == am_ref.item() == ref.abs().max().item()
    full = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double().reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2), padding=1)
    full = torch.nn.functional.leaky_relu(full * demod.double()[:, :, None, None] + 0.3 * noise.double().view(1, 1, H, H) + bias.double()[None, :, None, None], 0.2) * SQRT2
    assert (got.double().permute(0, 3, 1, 2) - full).abs().max() <= 2e-3 * full.abs().max()


@pytest.mark.parametrize('bm', [128, 256])
def test_repeated_launches_return_the_same_bits(dev, bm):
    """the counted waits of the DMA ring: a stale stage would show as run-to-run differences (as the mixed register / DMA form once did)"""
    B, Ci, Co, H = 32, 128, 128, 128
    x, plane, w, demod, noise, bias, epi = _case(dev, B, Ci, Co, H, 5)
    try:
        _flags(WGS_PATCH_DMA_BM=bm)
        first, _, sym = _run(plane, w, epi, B, H, Co, dev)
        assert sym.startswith('patch_dma_kernel<%d' % bm), sym
        busy = torch.randn(4096, 4096, device=dev)
        for i in range(20):
            if i % 2:
                busy = busy * 1.0001           # other traffic between the launches
            again, _, _ = _run(plane, w, epi, B, H, Co, dev)
            assert torch.equal(first, again), i
    finally:
        _flags(WGS_PATCH_DMA_BM=None)


@pytest.mark.parametrize('B,Ci,Co,H', [(32, 256, 256, 64), (16, 128, 512, 64), (13, 64, 256, 64), (4, 256, 256, 128)])
def test_256_columns_per_workgroup(dev, B, Ci, Co, H):
    """Cout % 256 == 0, WGS_PATCH_DMA_BN256=1: the 8-wave 256 x 256 tile (not the default route: 2-3 % slower than the LDS-DMA kernel).  Same bits as
    the register-staged patch kernel's 256 x 256 tile (same order of the MFMA sums); within fp32 rounding of the LDS-DMA kernel (another order)."""
    x, plane, w, demod, noise, bias, epi = _case(dev, B, Ci, Co, H, B + Ci + Co)
    try:
        _flags(WGS_PATCH_DMA_BN256=1)
        got, am, sym = _run(plane, w, epi, B, H, Co, dev)
        assert sym.startswith('patch_dma_kernel<256, 256, false>'), sym
        _flags(WGS_PATCH_DMA_BN256=None, WGS_PATCH_NODMA=1, WGS_PLANE_PATCH_MAX_CO=100000)
        ref, am_ref, sym_ref = _run(plane, w, epi, B, H, Co, dev)
        assert sym_ref.startswith('igemm_patch_kernel<1, 256, 256'), sym_ref
        _flags(WGS_PATCH_NODMA=None, WGS_PLANE_PATCH_MAX_CO=None)
        dma, _, sym_dma = _run(plane, w, epi, B, H, Co, dev)
        assert sym_dma.startswith('igemm_dma16_kernel'), sym_dma
    finally:
        _flags(WGS_PATCH_NODMA=None, WGS_PLANE_PATCH_MAX_CO=None, WGS_PATCH_DMA_BN256=None)
    assert torch.equal(got, ref) and am.item() == am_ref.item()
    assert (got - dma).abs().max() <= 2e-6 * dma.abs().max()
    full = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double().reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2), padding=1)
    full = torch.nn.functional.leaky_relu(full * demod.double()[:, :, None, None] + 0.3 * noise.double().view(1, 1, H, H) + bias.double()[None, :, None, None], 0.2) * SQRT2
    assert (got.double().permute(0, 3, 1, 2) - full).abs().max() <= 2e-3 * full.abs().max()


@pytest.mark.parametrize('B,Ci,Co,H', [(1, 128, 128, 64), (8, 128, 256, 64)])
def test_declined_shapes_run_through_the_other_kernels(dev, B, Ci, Co, H):
    x, plane, w, demod, noise, bias, epi = _case(dev, B, Ci, Co, H, 3)
    got, _, sym = _run(plane, w, epi, B, H, Co, dev)
    assert not sym.startswith('patch_dma_kernel'), sym
    full = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double().reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2), padding=1)
    full = torch.nn.functional.leaky_relu(full * demod.double()[:, :, None, None] + 0.3 * noise.double().view(1, 1, H, H) + bias.double()[None, :, None, None], 0.2) * SQRT2
    assert (got.double().permute(0, 3, 1, 2) - full).abs().max() <= 2e-3 * full.abs().max()
