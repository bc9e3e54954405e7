"""GPU: HIP ProgGAN generator vs the reference golden (full 1024^2 network) and the oracle."""
import pytest
import torch
import torch.nn.functional as F

from oracle import wgs_oracle as O
from tests import golden_inputs as GI
from tests.util import rel_err
from warpedganspace_amd.proggan import Generator, ProgGANWrapper

pytestmark = pytest.mark.gpu


def test_proggan_1024_vs_reference_golden(dev, golden):
    g = golden('generators')
    G = Generator()
    G.load_state_dict(GI.fill_state_dict(G.state_dict(), 500))
    wrap = ProgGANWrapper(G).to(dev).eval()
    z = GI.rt(501, 2, 512).to(dev)
    sh = (GI.rt(502, 2, 512) * 0.1).to(dev).requires_grad_(True)
    img = wrap(z, sh)
    assert img.shape == (2, 3, 1024, 1024)
    (F.avg_pool2d(img, 32) * GI.rt(503, 2, 3, 32, 32).to(dev)).sum().backward()
    assert rel_err(F.avg_pool2d(img.detach(), 32), g['proggan_img_pool32']) < 1e-4
    assert rel_err(img.detach()[:, :, 500:516, 300:316], g['proggan_img_crop']) < 1e-4
    e = rel_err(sh.grad, g['proggan_dshift'])
    print('ProgGAN-1024 d/dshift vs reference fp32: %.3e' % e)
    assert e < 5e-3      # loose envelope (leaky-relu gate flips between two fp32 evaluations); exact check below


@pytest.mark.parametrize('nb,B', [(8, 3), (12, 2)])
def test_proggan_truncated_vs_oracle_fp64(dev, nb, B):
    """Truncated networks (8 blocks = 32x32, 12 blocks = 128x128): image and input gradient vs the oracle in
    float64, the oracle differentiating through the SAME leaky-relu gates as the HIP forward (exact check)."""
    G = Generator(nb)
    sd = GI.fill_state_dict(G.state_dict(), 600 + nb)
    G.load_state_dict(sd)
    sd64 = {k: v.double() for k, v in sd.items()}
    z = GI.rt(601, B, 512)
    G.debug_keep = {}
    shd = (GI.rt(602, B, 512) * 0.2).to(dev).requires_grad_(True)
    img = ProgGANWrapper(G).to(dev)(z.to(dev), shd)
    probe = GI.rt(603, *img.shape)
    (img * probe.to(dev)).sum().backward()
    sh = (GI.rt(602, B, 512) * 0.2).double().requires_grad_(True)
    O.GATE_OVERRIDE = iter([g.cpu() for g in G.debug_keep['gates']])
    img_o = O.proggan_generate(sd64, z.double(), sh, num_blocks=nb)
    O.GATE_OVERRIDE = None
    (img_o * probe.double()).sum().backward()
    assert rel_err(img, img_o.detach()) < 1e-4
    e = rel_err(shd.grad, sh.grad)
    print('ProgGAN %d blocks: shared-gate d/dshift vs fp64 oracle %.3e' % (nb, e))
    assert e < 1e-4


@pytest.mark.parametrize('pauses', [16, (8, 32), (4, 8, 16, 32, 64), 256])
def test_staged_pass_equals_the_plain_pass_and_hooks_fire_in_the_backward(dev, pauses):
    """ProgGANWrapper.begin / advance / finish (the pass as a generator that pauses above the given resolutions; trainer.TrainStep runs the
    stages at different points of a training step) enqueue exactly the launches of a plain call: bit-identical image.  Generator.bwd_hooks:
    each (resolution, callable) is called once, largest resolution first, and the gradient is what it is without hooks."""
    G = Generator(10)                      # 64 x 64
    G.load_state_dict(GI.fill_state_dict(G.state_dict(), 700))
    wrap = ProgGANWrapper(G).to(dev).eval()
    z = GI.rt(701, 3, 512).to(dev)
    with torch.no_grad():
        ref = wrap(z)
        h = wrap.begin(z, pause_res=pauses)
        n, img = 0, None
        while img is None:
            img = wrap.advance(h)
            n += 1
    assert torch.equal(img, ref)
    expect = {16: 1, (8, 32): 2, (4, 8, 16, 32, 64): 4, 256: 1}[pauses]      # pauses above 64 never trigger; 64 x 64 is the last block's size
    assert n == expect, n
    sh = (GI.rt(702, 3, 512) * 0.1).to(dev)
    wgt = GI.rt(703, 3, 3, 64, 64).to(dev)
    grads, fired = [], []
    for hooks in (None, [(8, lambda: fired.append(8)), (32, lambda: fired.append(32)), (1024, lambda: fired.append(1024))]):
        s = sh.clone().requires_grad_(True)
        G.bwd_hooks = hooks
        (wrap(z, s) * wgt).sum().backward()
        assert G.bwd_hooks is None
        grads.append(s.grad.clone())
    assert fired == [1024, 32, 8]
    assert rel_err(grads[1], grads[0]) < 1e-5          # (atomic partial sums: the order of additions is not fixed)


def test_f16_backward_uses_the_magnitude_chain_and_tracks_fp32(dev):
    """In the fp16 modes every PixelNorm backward publishes max |dpre| (wgs_pixelnorm_bwd_act_amax) and the gradient conv behind it rounds
    dpre to fp16 under that bound (before: split-bf16 for want of a bound).  The bound is the tensor's true maximum; the input gradient
    stays with the exact-fp32 one (different leaky-relu gates on a few pre-activations: direction and size, not bits)."""
    G = Generator(12)                      # 128 x 128
    G.load_state_dict(GI.fill_state_dict(G.state_dict(), 800))
    wrap = ProgGANWrapper(G).to(dev).eval()
    z = GI.rt(801, 4, 512).to(dev)
    sh = (GI.rt(802, 4, 512) * 0.1).to(dev)
    wgt = GI.rt(803, 4, 3, 128, 128).to(dev)
    from warpedganspace_amd import proggan as PG
    grads = {}
    try:
        for prec, chain in (('fp32', True), ('bf16x3', True), ('f16', True), ('f16', False), ('f16x2', True), ('f16x2', False)):
            PG.F16_GRADS = chain
            s = sh.clone().requires_grad_(True)
            (wrap(z, s, precision=prec) * wgt).sum().backward()
            grads[(prec, chain)] = s.grad.double().flatten()
    finally:
        PG.F16_GRADS = True
    ref = grads[('fp32', True)]
    for prec in ('bf16x3', 'f16', 'f16x2'):
        g = grads[(prec, True)]
        cos = float((g * ref).sum() / (g.norm() * ref.norm()))
        err = float((g - ref).norm() / ref.norm())
        print('ProgGAN-128 d/dshift %s vs fp32: cosine %.6f, l2 error %.2e' % (prec, cos, err))
        # (another forward arithmetic = other leaky-relu gates on a few of the 1e7 pre-activations: direction and size, not digits)
        assert cos > (0.9999 if prec == 'bf16x3' else 0.995) and err < (2e-2 if prec == 'bf16x3' else 0.1), (prec, cos, err)
    for prec in ('f16', 'f16x2'):
        # the SAME forward (same gates), gradient convs in fp16 under the published bounds against split-bf16: the new path's own error
        g, g3 = grads[(prec, True)], grads[(prec, False)]
        err = float((g - g3).norm() / g3.norm())
        print('ProgGAN-128 d/dshift %s: fp16 gradient convs vs split-bf16 ones, same forward: l2 error %.2e' % (prec, err))
        assert 0 < err < 3e-3, (prec, err)
    # the published bound is the maximum of the tensor
    x = GI.rt(804, 2, 16, 16, 64).to(dev).contiguous()
    gy = GI.rt(805, 2, 16, 16, 64).to(dev).contiguous()
    am = torch.zeros(1, device=dev)
    gx = G._pixelnorm_bwd(x, gy, act_slope=0.2, amax=am)
    assert torch.equal(gx, G._pixelnorm_bwd(x, gy, act_slope=0.2))
    assert float(am) == float(gx.abs().max())
    x2, gy2 = GI.rt(806, 3, 1, 1, 512).to(dev).contiguous(), GI.rt(807, 3, 1, 1, 512).to(dev).contiguous()      # the scalar kernel (d = 512)
    am2 = torch.zeros(1, device=dev)
    gx2 = G._pixelnorm_bwd(x2, gy2, act_slope=1.0, amax=am2)
    assert float(am2) == float(gx2.abs().max())


@pytest.mark.parametrize('ci,co,H,up', [(16, 16, 256, False), (32, 16, 128, True), (32, 32, 256, False)])
@pytest.mark.parametrize('prec', ['bf16x3', 'f16', 'f16x2'])
def test_pixelnorm_inside_the_few_channel_kernel_is_bit_identical(dev, ci, co, H, up, prec):
    """wgs_conv_desc.a_pixelnorm_eps: the conv's operand PixelNorm(x) formed while conv_halo16.hip stages its input patch gives the bits of
    wgs_pixelnorm_fwd followed by the conv (same fma chain, same lane tree, same rounded product) — and the generator takes the route."""
    from warpedganspace_amd import conv as C
    from warpedganspace_amd import _lib as L
    B = 8
    x = (GI.rt(900 + ci, B, H, H, ci) * 3.0).to(dev).contiguous()
    w = (GI.rt(901 + co, co, 9, ci) / (9 * ci) ** 0.5).to(dev).contiguous()
    bias = GI.rt(902, co).to(dev)
    Ho = 2 * H if up else H
    taps = [(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
    m = C.precision_code(prec)
    kw = dict(w_tap_stride=ci, w_row_stride=9 * ci, ups=1 if up else 0, alpha=0.7, bias=bias, act_slope=0.2, gain=1.0, w_split=C.SplitCache(w), precision=m)
    y0, y1 = torch.empty(B, Ho, Ho, co, device=dev), torch.empty(B, Ho, Ho, co, device=dev)
    assert C.pixelnorm_fused_ok(x, w, y1, taps, Ho, Ho, **kw)
    xn = Generator._pixelnorm(x)
    lib = L.lib()
    lib.wgs_dev_trace_kernels(1)
    try:
        C.launch(xn, w, y0, taps, Ho, Ho, **kw)
        s0 = lib.wgs_dev_last_kernel().decode()
        C.launch(x, w, y1, taps, Ho, Ho, pixelnorm_eps=1e-8, **kw)
        s1 = lib.wgs_dev_last_kernel().decode()
    finally:
        lib.wgs_dev_trace_kernels(0)
    assert s0 == s1 and s0.startswith('halo3x3_kernel'), (s0, s1)
    assert torch.equal(y0, y1)
    # shapes the kernel does not cover are refused by the query and by the launch
    x64 = torch.randn(2, 64, 64, 64, device=dev)
    w64 = torch.randn(64, 9, 64, device=dev)
    y64 = torch.empty(2, 64, 64, 64, device=dev)
    kw64 = dict(w_tap_stride=64, w_row_stride=9 * 64, w_split=C.SplitCache(w64), precision=m)
    assert not C.pixelnorm_fused_ok(x64, w64, y64, taps, 64, 64, **kw64)
    with pytest.raises(L.WgsError):
        C.launch(x64, w64, y64, taps, 64, 64, pixelnorm_eps=1e-8, **kw64)


def test_generator_with_and_without_the_fused_pixelnorm(dev):
    from warpedganspace_amd import proggan as PG
    G = Generator(18)
    G.load_state_dict(GI.fill_state_dict(G.state_dict(), 950))
    wrap = ProgGANWrapper(G).to(dev).eval()
    z = GI.rt(951, 2, 512).to(dev)
    sh = (GI.rt(952, 2, 512) * 0.1).to(dev)
    out = {}
    try:
        for fused in (True, False):
            PG.PN_FUSED = fused
            G._pn_fused = {}
            s = sh.clone().requires_grad_(True)
            img = wrap(z, s, precision='f16')
            img.square().mean().backward()
            out[fused] = (img.detach().clone(), s.grad.clone(), sum(bool(v) for v in G._pn_fused.values()))
    finally:
        PG.PN_FUSED = True
    assert out[True][2] == 3 and out[False][2] == 0        # the 32 -> 32 @512^2, 32 -> 16 @1024^2 and 16 -> 16 @1024^2 blocks
    assert torch.equal(out[True][0], out[False][0])
    assert rel_err(out[True][1], out[False][1]) < 1e-5
