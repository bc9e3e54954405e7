"""GPU: HIP StyleGAN2 generator (forward + input gradient) vs the reference golden vectors and the oracle."""
import pytest
import torch
import torch.nn.functional as F

from oracle import wgs_oracle as O
from tests import golden_inputs as GI
from tests.util import rel_err, l2_rel
from warpedganspace_amd import _lib as L
from warpedganspace_amd.gan_load import StyleGAN2Wrapper
from warpedganspace_amd.stylegan2 import Generator

pytestmark = pytest.mark.gpu
TOL = 1e-3   # north_star: within 1e-3 relative fp32 of the reference path (measured values are ~1e-5)


def _check_grad(mine, ref32, ref64):
    """Gradient vs the reference golden (fp32) and vs the reference modules run in float64.  Differentiating
    through ~1e7..1e8 leaky-relu gates is only piecewise smooth: any two fp32 evaluations put a few
    pre-activations on opposite sides of zero, which moves the gradient by 1e-4..1e-3 (the reference's own
    fp32 result is 3e-4..1e-3 from its float64 self at 256^2; in Z space the random 8-layer mapping net
    amplifies it further).  So this check is a loose envelope; the EXACT check of the hand-derived backward
    is test_backward_exact_with_shared_gates (oracle forced through the HIP forward's gates: ~2e-6)."""
    e_ref = rel_err(ref32, ref64)
    e_mine = rel_err(mine, ref64)
    print('grad err vs fp64: hip %.3e, reference fp32 %.3e; hip vs reference fp32 %.3e' % (e_mine, e_ref, rel_err(mine, ref32)))
    assert e_mine < max(3 * e_ref, 3e-3), (e_mine, e_ref)
    assert rel_err(mine, ref32) < 5 * TOL


def build(size, seed, dev):
    G = Generator(size, 512, 8)
    sd = GI.fill_state_dict(G.state_dict(), seed)
    G.load_state_dict(sd)
    return G.to(dev), sd


@pytest.mark.parametrize('size', [32, 256])
def test_generator_vs_reference_golden(dev, golden, size):
    g = golden('stylegan2')
    G, sd = build(size, 400 + size, dev)
    tag = 'g%d_' % size
    z = GI.rt(410 + size, 2, 512).to(dev)
    shift = (GI.rt(411 + size, 2, 512) * 0.02).to(dev).requires_grad_(True)
    wrap = StyleGAN2Wrapper(G, shift_in_w_space=False)
    img = wrap(z, shift)
    probe = GI.rt(412 + size, *img.shape).to(dev)
    (img * probe).sum().backward()
    w = wrap.get_w(z)
    assert rel_err(w, g[tag + 'w']) < 1e-5
    if size == 32:
        e = rel_err(img, g[tag + 'img'])
    else:
        e = max(rel_err(F.avg_pool2d(img.detach(), 8), g[tag + 'img_pool8']),
                rel_err(img.detach()[:, :, 100:116, 60:76], g[tag + 'img_crop']))
    assert e < 1e-4, e
    _check_grad(shift.grad, g[tag + 'dshift'], g[tag + 'dshift64'])
    # W space
    wrapw = StyleGAN2Wrapper(G, shift_in_w_space=True)
    shw = (GI.rt(413 + size, 2, 512) * 0.05).to(dev).requires_grad_(True)
    imgw = wrapw(z, shw)
    (imgw * probe).sum().backward()
    if size == 32:
        assert rel_err(imgw, g[tag + 'w_img']) < 1e-4
    else:
        assert rel_err(F.avg_pool2d(imgw.detach(), 8), g[tag + 'w_img_pool8']) < 1e-4
    _check_grad(shw.grad, g[tag + 'w_dshift'], g[tag + 'w_dshift64'])
    # latent_is_w path (traverse_latent_space.py:457-462)
    imgw2 = wrapw(w.detach(), shw.detach(), latent_is_w=True)
    assert rel_err(imgw2, imgw) < 1e-6


def test_generator_vs_oracle_64_batch5(dev):
    """Another size / batch against the CPU oracle (full image; gradients judged against the oracle run
    in float64, because differentiating through ~1e7 leaky-relu gates makes ANY fp32 evaluation differ
    from the exact gradient at the 1e-4..1e-3 level: a few pre-activations round to the other sign)."""
    G, sd = build(64, 777, dev)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    z = GI.rt(778, 5, 512)
    probe = None
    grads = {}
    for name, dd, dt in (('f32', sd, torch.float32), ('f64', sd64, torch.float64)):
        shift = (GI.rt(779, 5, 512) * 0.3).to(dt).requires_grad_(True)
        img_o = O.sg2_generate(dd, z.to(dt), 64, shift)
        probe = GI.rt(780, *img_o.shape)
        (img_o * probe.to(dt)).sum().backward()
        w = O.sg2_mapping(dd, z.to(dt)).detach()
        shw = (GI.rt(781, 5, 512) * 0.1).to(dt).requires_grad_(True)
        (O.sg2_synthesis(dd, w + shw, 64) * probe.to(dt)).sum().backward()
        grads[name] = (img_o.detach(), shift.grad, shw.grad)
    sh = (GI.rt(779, 5, 512) * 0.3).to(dev).requires_grad_(True)
    img = StyleGAN2Wrapper(G, False)(z.to(dev), sh)
    (img * probe.to(dev)).sum().backward()
    shd = (GI.rt(781, 5, 512) * 0.1).to(dev).requires_grad_(True)
    (StyleGAN2Wrapper(G, True)(z.to(dev), shd) * probe.to(dev)).sum().backward()
    print('image err vs fp64 oracle: hip %.3e, fp32 oracle %.3e' % (rel_err(img, grads['f64'][0]),
                                                                  rel_err(grads['f32'][0], grads['f64'][0])))
    assert rel_err(img, grads['f64'][0]) < 1e-4
    for mine, i, what in ((sh.grad, 1, 'Z'), (shd.grad, 2, 'W')):
        e_ref = rel_err(grads['f32'][i], grads['f64'][i])
        e_mine = rel_err(mine, grads['f64'][i])
        print('%s-space grad err vs fp64 oracle: hip %.3e, fp32 oracle %.3e' % (what, e_mine, e_ref))
        assert e_mine < 3 * e_ref + 2e-4, (what, e_mine, e_ref)


@pytest.mark.parametrize('size,B', [(32, 3), (128, 2)])
def test_backward_exact_with_shared_gates(dev, size, B):
    """The hand-derived backward against autograd of the oracle in float64, with the oracle forced through
    the SAME leaky-relu gates the HIP forward took: every remaining difference is fp32 round-off."""
    G, sd = build(size, 31 + size, dev)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    z = GI.rt(32 + size, B, 512)
    probe = None
    for w_space in (False, True):
        G.debug_keep = {}
        sh = (GI.rt(33 + size, B, 512) * 0.1).to(dev).requires_grad_(True)
        img = StyleGAN2Wrapper(G, w_space)(z.to(dev), sh)
        probe = GI.rt(34 + size, *img.shape)
        (img * probe.to(dev)).sum().backward()
        gates = ([] if w_space else [g.cpu() for g in G.debug_keep['mapping']]) + [g.cpu() for g in G.debug_keep['synthesis']]
        G.debug_keep = None
        sho = (GI.rt(33 + size, B, 512) * 0.1).double().requires_grad_(True)
        if w_space:
            w = O.sg2_mapping(sd64, z.double()).detach()
            O.GATE_OVERRIDE = iter(gates)
            img_o = O.sg2_synthesis(sd64, w + sho, size)
        else:
            O.GATE_OVERRIDE = iter(gates)
            img_o = O.sg2_generate(sd64, z.double(), size, sho)
        O.GATE_OVERRIDE = None
        (img_o * probe.double()).sum().backward()
        e = rel_err(sh.grad, sho.grad)
        print('%s-space size %d: shared-gate gradient rel err %.3e' % ('W' if w_space else 'Z', size, e))
        assert rel_err(img, img_o.detach()) < 1e-4
        assert e < (2e-4 if w_space else 2e-3), e      # Z adds the ill-conditioned random mapping net


def test_no_grad_forward_saves_nothing(dev):
    G, _ = build(32, 5, dev)
    with torch.no_grad():
        img = StyleGAN2Wrapper(G, False)(torch.randn(3, 512, device=dev))
    assert img.shape == (3, 3, 32, 32) and not img.requires_grad


def test_linear_fwd_batch_equals_per_layer(dev):
    """wgs_linear_fwd_batch (all demodulation GEMVs of a pass in one launch) == one wgs_linear_fwd per layer, bit for bit."""
    import ctypes
    from warpedganspace_amd import _lib as L
    from warpedganspace_amd.stylegan2 import LinearBatch
    torch.manual_seed(3)
    B, sumC = 5, 512 + 256 + 64
    S = torch.randn(B, sumC, device=dev)
    layers = [(512, 512, 0, 0.02), (256, 512, 0, 0.03), (128, 256, 512, 0.05), (64, 64, 768, 0.07)]     # (Co, Ci, offset, scale)
    wsq = [torch.rand(co, ci, device=dev) for co, ci, _, _ in layers]
    out_b = [torch.empty(B, co, device=dev) for co, _, _, _ in layers]
    out_s = [torch.empty(B, co, device=dev) for co, _, _, _ in layers]
    lb = LinearBatch()
    lb.n, lb.M, lb.in_square, lb.epilogue = len(layers), B, 1, 2
    for i, (co, ci, off, sc) in enumerate(layers):
        lb.x[i] = S.data_ptr() + 4 * off
        lb.w[i], lb.y[i] = wsq[i].data_ptr(), out_b[i].data_ptr()
        lb.N[i], lb.K[i], lb.ldx[i], lb.ldy[i] = co, ci, sumC, co
        lb.wscale[i], lb.eps[i], lb.out_gain[i] = sc ** 2, 1e-8, sc
    L.check(L.lib().wgs_linear_fwd_batch(ctypes.byref(lb), L.stream()), 'batch')
    for i, (co, ci, off, sc) in enumerate(layers):
        L.check(L.lib().wgs_linear_fwd(L.rawptr(S[:, off:]), L.ptr(wsq[i]), None, L.ptr(out_s[i]), B, co, ci, sumC, co,
                                       L.c_float(sc ** 2), L.c_float(0.0), 1, 2, L.c_float(1e-8), L.c_float(sc), L.stream()), 'single')
        assert torch.equal(out_b[i], out_s[i])
        ref = sc / torch.sqrt(sc ** 2 * (S[:, off:off + ci].double() ** 2 @ wsq[i].double().t()) + 1e-8)
        assert (out_b[i].double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize('size,B', [(512, 2), (1024, 1)])
def test_high_resolution_architectures(dev, size, B):
    """The 512 / 1024 architectures (cfg5: channels down to 64 / 32 — the narrow-Cout tiles, Cin = 32 chunks, ToRGB on 32
    channels): forward image against the CPU oracle, split-bf16 against the exact-fp32 kernels, and the input gradient of
    the two arithmetic modes against each other."""
    G, sd = build(size, 4242 + size, dev)
    z = GI.rt(11 + size, B, 512)
    shift = GI.rt(12 + size, B, 512) * 0.3
    with torch.no_grad():
        ref = O.sg2_generate(sd, z, size, shift)
    outs, grads = {}, {}
    for prec in (0, 1):
        sh = shift.to(dev).requires_grad_(True)
        img = StyleGAN2Wrapper(G, False)(z.to(dev), sh, precision=prec)
        probe = GI.rt(13 + size, *img.shape).to(dev)
        (img * probe).sum().backward()
        outs[prec], grads[prec] = img.detach(), sh.grad.detach()
    e0, e1 = rel_err(outs[0], ref), rel_err(outs[1], ref)
    print('StyleGAN2-%d: image vs oracle: exact fp32 %.2e, split-bf16 %.2e; grad split vs exact %.2e' % (size, e0, e1, rel_err(grads[1], grads[0])))
    assert e0 < 1e-4 and e1 < 1e-4                      # gate of the north_star: 1e-3
    assert rel_err(outs[1], outs[0]) < 1e-4
    assert rel_err(grads[1], grads[0]) < 5e-2           # free-running gradients: leaky-relu gate flips (see above)


def test_full_size_batch_consistency(dev):
    """BASELINE-size property (256x256, batch 32, both arithmetic modes): a sample's image and input gradient do not depend on
    what else is in the batch.  The batch-32 run takes the large-tile / merged-phase / patch / LDS-DMA kernels, the batch-2
    run mostly the 128-row and split-K ones, so this also cross-checks the kernel families against each other."""
    G, _ = build(256, 909, dev)
    z = GI.rt(910, 32, 512).to(dev)
    shift = (GI.rt(911, 32, 512) * 0.3).to(dev)
    probe = GI.rt(912, 32, 3, 256, 256).to(dev)
    wrap = StyleGAN2Wrapper(G, False)
    for prec, tol in ((1, 2e-5), (0, 2e-5)):
        sh = shift.clone().requires_grad_(True)
        img = wrap(z, sh, precision=prec)
        (img * probe).sum().backward()
        for sl in (slice(0, 2), slice(30, 32)):
            sh2 = shift[sl].clone().requires_grad_(True)
            img2 = wrap(z[sl], sh2, precision=prec)
            (img2 * probe[sl]).sum().backward()
            e_img, e_g = rel_err(img2, img[sl]), rel_err(sh2.grad, sh.grad[sl])
            print('precision %d rows %s: image %.2e grad %.2e' % (prec, sl, e_img, e_g))
            assert e_img < tol
            assert e_g < 5e-2              # free-running gradient: gate flips between two fp32-class evaluations


@pytest.mark.parametrize('pauses', [32, (8, 32), (16, 128), 1024])
@pytest.mark.parametrize('prec', ['fp32', 'auto'])
def test_staged_pass_equals_the_plain_pass_and_hooks_fire_in_the_backward(dev, pauses, prec):
    """StyleGAN2Wrapper.begin / advance / finish (the un-shifted pass as a generator that pauses above the given resolutions: trainer.TrainStep
    runs its stages at different points of a training step) enqueue exactly the launches of a plain call: bit-identical image.
    Generator.bwd_hooks: each (resolution, callable) is called once, largest resolution first; the gradient is what it is without hooks."""
    G, _ = build(64, 41, dev)
    wrap = StyleGAN2Wrapper(G, False).eval()
    z = GI.rt(42, 4, 512).to(dev)
    with torch.no_grad():
        ref = wrap(z, precision=prec)
        h = wrap.begin(z, precision=prec, pause_res=pauses)
        n, img = 0, None
        while img is None:
            img = wrap.advance(h)
            n += 1
        assert torch.equal(wrap.finish(wrap.begin(z, precision=prec, pause_res=pauses)), ref)
    assert torch.equal(img, ref)
    assert n == {32: 1, (8, 32): 2, (16, 128): 1, 1024: 1}[pauses], n
    sh = (GI.rt(43, 4, 512) * 0.1).to(dev)
    wgt = GI.rt(44, 4, 3, 64, 64).to(dev)
    grads, fired = [], []
    for hooks in (None, [(8, lambda: fired.append(8)), (32, lambda: fired.append(32)), (4096, lambda: fired.append(4096))]):
        s = sh.clone().requires_grad_(True)
        G.bwd_hooks = hooks
        (wrap(z, s, precision=prec) * wgt).sum().backward()
        assert G.bwd_hooks is None
        grads.append(s.grad.clone())
    assert fired == [4096, 32, 8]
    assert rel_err(grads[1], grads[0]) < 1e-5          # (atomic partial sums: the order of additions is not fixed)


def test_stylegan2_1024_torgb_in_the_few_channel_kernel(dev):
    """StyleGAN2-1024 under its default 'mixed' table: the ToRGBs of the 64- / 32-channel layers at 512^2 / 1024^2 run in their conv's epilogue
    (conv.rgb_halo_ok); same image and gradient as with the separate ToRGB launches."""
    from warpedganspace_amd import conv as C
    G, _ = build(1024, 77, dev)
    wrap = StyleGAN2Wrapper(G, False).eval()
    z = GI.rt(78, 2, 512).to(dev)
    sh = (GI.rt(79, 2, 512) * 0.1).to(dev)
    out = {}
    try:
        for fused in (True, False):
            C.RGB_FUSED = fused
            G._route = {}
            s = sh.clone().requires_grad_(True)
            img = wrap(z, s, precision='mixed')
            img.square().mean().backward()
            with torch.no_grad():
                img0 = wrap(z, precision='mixed')             # the pass that keeps nothing (its last layer's output is never stored)
            out[fused] = (img.detach().clone(), s.grad.clone(), img0.clone(), sum(bool(v) for k, v in G._route.items() if k[0] == 'rgb_halo'))
    finally:
        C.RGB_FUSED = True
    assert out[True][3] >= 2 and out[False][3] == 0
    assert rel_err(out[True][0], out[False][0]) < 1e-5 and rel_err(out[True][2], out[False][2]) < 1e-5
    assert rel_err(out[True][1], out[False][1]) < 1e-4


def test_backward_hooks_belong_to_one_forward_and_stages_to_one_stream(dev):
    """the hook list set before a differentiable forward is bound to THAT forward's autograd node — entries appended after the
    forward fire in its backward, and another forward / backward of the same generator in between neither sees nor consumes them.  A
    staged pass captured its launch stream at begin(): resuming it under another stream raises instead of enqueueing there."""
    G, _ = build(32, 41, dev)
    wrap = StyleGAN2Wrapper(G, False).eval()
    z = GI.rt(42, 2, 512).to(dev)
    wgt = GI.rt(44, 2, 3, 32, 32).to(dev)
    fired, carrier = [], []
    s1 = (GI.rt(43, 2, 512) * 0.1).to(dev).requires_grad_(True)
    G.bwd_hooks = carrier
    out1 = wrap(z, s1)
    assert G.bwd_hooks is None                       # taken by the forward
    carrier.append((8, lambda: fired.append('mine')))
    s2 = (GI.rt(45, 2, 512) * 0.1).to(dev).requires_grad_(True)
    (wrap(z, s2) * wgt).sum().backward()             # an unrelated forward / backward of the same generator
    assert fired == [] and len(carrier) == 1
    (out1 * wgt).sum().backward()
    assert fired == ['mine'] and carrier == []
    with torch.no_grad():
        h = wrap.begin(z, pause_res=8)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            with pytest.raises(L.WgsError):
                wrap.advance(h)
        assert torch.equal(wrap.finish(h), wrap(z))
