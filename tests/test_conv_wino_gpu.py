"""GPU: the Winograd F(2x2, 3x3) fp32 conv kernel (csrc/conv_wino_f32.hip, precision 'fp32w') against float64 torch convs and
against the direct exact-fp32 kernel — the styled forward conv of models/StyleGAN2/model.py:187-228 (style, demodulation, noise,
bias, leaky-relu epilogue) and its input-gradient form (transposed weights, flipped taps)."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_err
from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C

pytestmark = pytest.mark.gpu


def _supported(x, w, y, **kw):
    d, _ = C._desc(x, w, y, [(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)], y.shape[1], y.shape[2],
                   w_tap_stride=x.shape[3], w_row_stride=9 * x.shape[3], **kw)
    import ctypes
    return bool(L.lib().wgs_conv_wino_supported(ctypes.byref(d)))


@pytest.mark.parametrize('B,H,W,ci,co', [(2, 16, 16, 64, 64), (3, 32, 16, 16, 128), (2, 48, 64, 128, 64), (1, 64, 64, 512, 512), (5, 16, 32, 48, 192)])
def test_wino_styled_forward_vs_float64(dev, B, H, W, ci, co):
    torch.manual_seed(B * 1000 + ci + co)
    x = torch.randn(B, H, W, ci)
    w = torch.randn(co, ci, 3, 3) / (9 * ci) ** 0.5
    s = torch.randn(B, ci) + 1.0
    dm = torch.rand(B, co) + 0.5
    bias, noise, nw = torch.randn(co), torch.randn(H * W), torch.tensor([0.37])
    xd = (x.double() * s.double()[:, None, None, :]).permute(0, 3, 1, 2)
    ref = F.conv2d(xd, w.double(), padding=1) * dm.double()[:, :, None, None]
    ref = ref + (nw.double() * noise.double()).reshape(1, 1, H, W) + bias.double()[None, :, None, None]
    ref = (F.leaky_relu(ref, 0.2) * 2 ** 0.5).permute(0, 2, 3, 1)
    wp = C.pack_weight(w.to(dev))
    xg = x.to(dev)
    epi = dict(a_scale=s.to(dev), col_scale=dm.to(dev), bias=bias.to(dev), noise=noise.to(dev), noise_w=nw.to(dev), act_slope=0.2, gain=2 ** 0.5)
    y = torch.empty(B, H, W, co, device=dev)
    assert _supported(xg, wp, y, **epi)
    L.lib().wgs_dev_trace_kernels(1)
    ymax = torch.zeros(1, device=dev)
    got = C.conv2d(xg, wp, 3, pad=1, precision=C.FP32W, y_amax=ymax, **epi)
    assert float(ymax) == float(got.abs().max())           # the magnitude scalar the fp16 chains read (exact: a maximum, not a sum)
    sym = L.lib().wgs_dev_last_kernel().decode()
    assert sym.startswith('wino_f32_kernel<') and ', true, ' in sym, sym
    direct = C.conv2d(xg, wp, 3, pad=1, precision=0, **epi)
    L.lib().wgs_dev_trace_kernels(0)
    e_w, e_d = rel_err(got, ref), rel_err(direct, ref)
    assert e_w < 3e-6, (e_w, e_d)
    assert rel_err(got, direct) < 4e-6


@pytest.mark.parametrize('B,H,ci,co', [(2, 32, 64, 128), (2, 16, 256, 64), (1, 48, 128, 256)])
def test_wino_input_gradient_form_vs_float64(dev, B, H, ci, co):
    """dgrad of a 3x3 stride-1 pad-1 conv = the same kernel on dy with the transposed packed weights and flipped taps."""
    torch.manual_seed(7 + ci)
    w = torch.randn(co, ci, 3, 3) / (9 * ci) ** 0.5
    dy = torch.randn(B, H, H, co)
    xd = torch.zeros(B, ci, H, H, dtype=torch.double, requires_grad=True)
    F.conv2d(xd, w.double(), padding=1).backward(dy.double().permute(0, 3, 1, 2))
    ref = xd.grad.permute(0, 2, 3, 1)
    wp = C.pack_weight(w.to(dev))
    wt = C.repack_w_t(wp, co, 9, ci)
    cache = C.SplitCache(wt)
    L.lib().wgs_dev_trace_kernels(1)
    got = C.conv2d_dgrad(dy.to(dev), wt, (H, H), 3, pad=1, precision=C.FP32W, w_split=cache, alpha=1.0)
    sym = L.lib().wgs_dev_last_kernel().decode()
    assert sym.startswith('wino_f32_kernel<') and ', false, ' in sym, sym
    L.lib().wgs_dev_trace_kernels(0)
    assert rel_err(got, ref) < 3e-6
    assert len(cache.planes) == 1           # U is kept with the weight tensor
    again = C.conv2d_dgrad(dy.to(dev), wt, (H, H), 3, pad=1, precision=C.FP32W, w_split=cache)
    assert torch.equal(got, again) and len(cache.planes) == 1


def test_wino_declines_what_it_does_not_cover_and_the_direct_kernel_runs(dev):
    torch.manual_seed(3)
    x = torch.randn(2, 8, 8, 64, device=dev)           # 8x8: not a multiple of the 16x16 pixel block
    wp = C.pack_weight(torch.randn(64, 64, 3, 3, device=dev) / 24)
    y = torch.empty(2, 8, 8, 64, device=dev)
    assert not _supported(x, wp, y)
    L.lib().wgs_dev_trace_kernels(1)
    got = C.conv2d(x, wp, 3, pad=1, precision=C.FP32W)
    assert 'wino' not in L.lib().wgs_dev_last_kernel().decode()
    L.lib().wgs_dev_trace_kernels(0)
    assert torch.equal(got, C.conv2d(x, wp, 3, pad=1, precision=0))
    x2 = torch.randn(2, 16, 16, 64, device=dev)
    y2 = torch.empty(2, 8, 8, 64, device=dev)
    assert not _supported(x2, wp, torch.empty(2, 16, 16, 64, device=dev), act=1)


def test_wino_wide_and_narrow_workgroup_shapes_agree(dev, monkeypatch):
    """Cout % 128 == 0 runs 32 tiles x 128 channels per workgroup, WGS_WINO_NARROW pins 64 x 64: same products, same sums per output."""
    torch.manual_seed(9)
    x = torch.randn(2, 32, 32, 64, device=dev)
    wp = C.pack_weight(torch.randn(128, 64, 3, 3, device=dev) / 24)
    s = torch.randn(2, 64, device=dev)
    lib = L.lib()
    lib.wgs_dev_trace_kernels(1)
    wide = C.conv2d(x, wp, 3, pad=1, precision=C.FP32W, a_scale=s)
    k_wide = lib.wgs_dev_last_kernel().decode()
    monkeypatch.setenv('WGS_WINO_NARROW', '1')
    lib.wgs_dev_reload_flags()
    try:
        narrow = C.conv2d(x, wp, 3, pad=1, precision=C.FP32W, a_scale=s)
        k_narrow = lib.wgs_dev_last_kernel().decode()
    finally:
        monkeypatch.delenv('WGS_WINO_NARROW')
        lib.wgs_dev_reload_flags()
        lib.wgs_dev_trace_kernels(0)
    assert k_wide == 'wino_f32_kernel<1, 2, true, 4>' and k_narrow == 'wino_f32_kernel<2, 2, true, 8>'      # 16 items: the small-grid shape
    assert rel_err(wide, narrow) < 2e-6
    xb = torch.randn(8, 64, 64, 64, device=dev)                # 256 items of the wide shape: 8 waves, 32 tiles x 128 channels
    sb_ = torch.randn(8, 64, device=dev)
    lib.wgs_dev_trace_kernels(1)
    big = C.conv2d(xb, wp, 3, pad=1, precision=C.FP32W, a_scale=sb_)
    assert lib.wgs_dev_last_kernel().decode() == 'wino_f32_kernel<1, 4, true, 8>'
    lib.wgs_dev_trace_kernels(0)
    assert rel_err(big, C.conv2d(xb, wp, 3, pad=1, precision=0, a_scale=sb_)) < 4e-6


def test_fp32w_through_the_other_generators_plain_3x3_layers(dev):
    """'fp32w' is a mode of every generator: ProgGAN's second conv of a block (plain 3x3, WScale alpha, bias, leaky-relu) is a launch
    the Winograd kernel covers, its first (nearest-neighbour up-sampled gather) is not — image and input gradient against 'fp32'."""
    from tests import golden_inputs as GI
    from warpedganspace_amd.proggan import Generator, ProgGANWrapper
    G = Generator(12)                               # 128 x 128
    G.load_state_dict(GI.fill_state_dict(G.state_dict(), 612))
    wrap = ProgGANWrapper(G).to(dev).eval()
    z = GI.rt(613, 2, 512).to(dev)
    out = {}
    lib = L.lib()
    for mode in ('fp32', 'fp32w'):
        sh = (GI.rt(614, 2, 512) * 0.1).to(dev).requires_grad_(True)
        img = wrap(z, sh, precision=mode)
        (img * GI.rt(615, *img.shape).to(dev)).sum().backward()
        out[mode] = (img.detach(), sh.grad.clone())
    assert rel_err(out['fp32w'][0], out['fp32'][0]) < 2e-5
    assert rel_err(out['fp32w'][1], out['fp32'][1]) < 5e-3           # the envelope test_proggan_gpu.py uses between two fp32 evaluations: leaky-relu gates flip
    lib.wgs_dev_trace_kernels(1)
    C.PROFILE = []
    try:
        with torch.no_grad():
            wrap(z, precision='fp32w')
        torch.cuda.synchronize()
        syms = {r[4] for r in C.PROFILE}
    finally:
        C.PROFILE = None
        lib.wgs_dev_trace_kernels(0)
    assert any(s.startswith('wino_f32_kernel') for s in syms), syms


def test_one_weight_cache_serves_launches_of_different_layouts(dev):
    """U's fragment order follows the workgroup shape, which depends on the batch (512 -> 512 @16^2 takes the 4-wave
    shape up to B = 24 and the 8-wave 128-channel shape from B = 25).  One SplitCache shared by a B = 1 and a B = 32 launch (what
    sample_gan.py's final partial batch does) must hand each its own layout: both results against fp64, and two cached operands."""
    import ctypes
    torch.manual_seed(3)
    ci = co = 512
    w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    cache = C.SplitCache(w)
    wr = w.double().reshape(co, 3, 3, ci).permute(0, 3, 1, 2)
    layouts = set()
    for B in (32, 1, 32, 2):
        x = torch.randn(B, 16, 16, ci, device=dev)
        y = C.conv2d(x, w, 3, pad=1, precision=C.FP32W, w_split=cache)
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wr, padding=1).permute(0, 2, 3, 1)
        e = float((y.double() - ref).abs().max() / ref.abs().max())
        assert e < 3e-6, (B, e)
        d, _ = C._desc(x, w, y, [(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)], 16, 16, w_tap_stride=ci, w_row_stride=9 * ci,
                       precision=C.FP32W)
        layouts.add(L.lib().wgs_conv_wino_layout(ctypes.byref(d)))
    assert len(layouts) == 2, layouts                                   # the two batch sizes really straddle the shape threshold
    assert sum(1 for k in cache.planes if isinstance(k, tuple) and k[0] == 'wino') == 2


def test_step_cache_follows_its_weight_tensor(dev):
    """a StepWinoCache recipe keeps the weight tensor it was recorded from.  refresh() rebuilds U from the tensor's CURRENT
    values; a launch whose weights live at another address by now (parameters re-homed, R.to(), a re-allocated transposed copy) does not
    hit the stale operand — it rebuilds from its own tensor and the recipe follows it."""
    torch.manual_seed(5)
    ci = co = 64
    x = torch.randn(2, 32, 32, ci, device=dev)

    def ref(w):
        return torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().reshape(co, 3, 3, ci).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)

    def err(y, w):
        r = ref(w)
        return float((y.double() - r).abs().max() / r.abs().max())
    cache = C.StepWinoCache()
    w1 = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    assert err(C.conv2d(x, w1, 3, pad=1, precision=C.FP32W, w_split=cache), w1) < 3e-6
    assert len(cache.recipes) == 1
    w1.mul_(-0.5)                                   # an optimiser update in place: stale until refresh() ...
    assert err(C.conv2d(x, w1, 3, pad=1, precision=C.FP32W, w_split=cache), w1) > 0.1
    cache.refresh()                                 # ... which re-reads the tensor
    assert err(C.conv2d(x, w1, 3, pad=1, precision=C.FP32W, w_split=cache), w1) < 3e-6
    w2 = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5          # the same layer's weights at ANOTHER address
    assert err(C.conv2d(x, w2, 3, pad=1, precision=C.FP32W, w_split=cache), w2) < 3e-6
    assert len(cache.recipes) == 1 and next(iter(cache.recipes.values()))[1] is w2
    del w1
    w2.add_(0.01)
    cache.refresh()
    assert err(C.conv2d(x, w2, 3, pad=1, precision=C.FP32W, w_split=cache), w2) < 3e-6
