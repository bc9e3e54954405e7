"""GPU: the split-bf16 (3 x bf16 MFMA) arithmetic mode of the conv kernels, end to end, against the same goldens /
oracle as the exact-fp32 mode, at the north_star's gate (1e-3 relative; path-index argmax bit-exact)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import wgs_oracle as O
from tests import golden_inputs as GI
from tests.util import rel_err
from warpedganspace_amd import conv as C
from warpedganspace_amd.gan_load import StyleGAN2Wrapper

pytestmark = pytest.mark.gpu


def test_stylegan2_256_forward_and_gradient(dev, golden):
    from tests.test_stylegan2_gpu import build
    g = golden('stylegan2')
    G, sd = build(256, 400 + 256, dev)
    z = GI.rt(410 + 256, 2, 512).to(dev)
    shift = (GI.rt(411 + 256, 2, 512) * 0.02).to(dev).requires_grad_(True)
    img = StyleGAN2Wrapper(G, False)(z, shift, precision='bf16x3')
    probe = GI.rt(412 + 256, *img.shape).to(dev)
    (img * probe).sum().backward()
    e = max(rel_err(F.avg_pool2d(img.detach(), 8), g['g256_img_pool8']), rel_err(img.detach()[:, :, 100:116, 60:76], g['g256_img_crop']))
    eg = rel_err(shift.grad, g['g256_dshift64'])
    print('split-bf16 StyleGAN2-256: image rel err %.2e, d/dshift vs fp64 reference %.2e' % (e, eg))
    assert e < 1e-3
    assert eg < 1e-2     # Z-space gradient through ~1e8 leaky-relu gates + random mapping net (see test_stylegan2_gpu)


def test_reconstructor_stays_exact_fp32(dev):
    """A Reconstructor on its own runs the reference's arithmetic (exact fp32 convs), whatever generators run in."""
    from tests.test_reconstructor_gpu import _run_pair
    R, sd, (lo, mo, x2), (lg, mg, x2d) = _run_pair(dev, 4, 128, 64)
    assert rel_err(lg, lo.detach()) < 1e-4 and rel_err(mg, mo.detach()) < 1e-4
    assert torch.equal(torch.argmax(lg, 1).cpu(), torch.argmax(lo, 1))
    assert max(rel_err(p.grad, sd[n].grad) for n, p in R.named_parameters() if p.grad is not None) < 1e-3


def test_training_step_loss_and_argmax(dev):
    from tests.test_train_step_gpu import make
    eng, ref, c = make(dev, 32, 16, 4, 4, False, precision='bf16x3')
    g = torch.Generator().manual_seed(7)
    z = torch.randn(4, 512, generator=g)
    idx = torch.randint(0, 16, (4,), generator=g)
    mag = (torch.rand(4, generator=g) * 0.2 + 0.25)
    o = ref.step(z, idx, mag)
    st = eng.step(z.to(dev), idx.to(dev), mag.to(dev)).tolist()
    assert abs(st[2] - o['loss']) < 1e-3 * max(1.0, abs(o['loss']))
    assert torch.equal(eng.argmax.cpu(), o['argmax'])
