"""GPU: HIP BigGAN-128 generator (ccbn, SN constants, self-attention) vs the reference golden and the oracle."""
import pytest
import torch
import torch.nn.functional as F

from oracle import wgs_oracle as O
from tests import golden_inputs as GI
from tests.test_oracle_golden import _biggan
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def test_biggan_vs_reference_golden_and_shared_gate_oracle(dev, golden):
    g = golden('generators')
    W = _biggan()
    sd = {k: v.detach().clone() for k, v in W.G.state_dict().items()}
    G = W.G.to(dev).eval()
    assert W.dim_z == 120
    z = GI.rt(541, 2, 120)
    cls = torch.tensor([239, 100])
    shd = (GI.rt(542, 2, 120) * 0.1).to(dev).requires_grad_(True)
    G.debug_keep = {}
    img = G(z.to(dev) + shd, G.shared(cls.to(dev)))
    assert img.shape == (2, 3, 128, 128)
    probe = GI.rt(543, *img.shape)
    (img * probe.to(dev)).sum().backward()
    assert rel_err(F.avg_pool2d(img.detach(), 4), g['biggan_img_pool4']) < 1e-4
    assert rel_err(img.detach()[:, :, 40:56, 70:86], g['biggan_img_crop']) < 1e-4
    e = rel_err(shd.grad, g['biggan_dshift'])
    print('BigGAN d/dshift vs reference fp32: %.3e' % e)
    assert e < 5e-3
    # exact check: oracle in float64 through the same ReLU gates
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    sh = (GI.rt(542, 2, 120) * 0.1).double().requires_grad_(True)
    O.GATE_OVERRIDE = iter([m.cpu() for m in G.debug_keep['gates']])
    img_o = O.biggan_generate(sd64, z.double(), cls, sh)
    O.GATE_OVERRIDE = None
    (img_o * probe.double()).sum().backward()
    assert rel_err(img, img_o.detach()) < 5e-4
    e2 = rel_err(shd.grad, sh.grad)
    print('BigGAN shared-gate d/dshift vs fp64 oracle: %.3e' % e2)
    assert e2 < 1e-3


def test_biggan_wrapper_draws_target_classes(dev):
    W = _biggan()
    W.target_classes.data = torch.tensor([14, 239])
    W = W.to(dev).eval()
    with torch.no_grad():
        img = W(torch.randn(3, 120, device=dev))
    assert img.shape == (3, 3, 128, 128) and float(img.abs().max()) <= 1.0


@pytest.mark.parametrize('pauses', [32, (16, 64), 512])
def test_staged_pass_equals_the_plain_pass_and_hooks_fire_in_the_backward(dev, pauses):
    """BigGANWrapper.begin / advance / finish (the pass as a generator that pauses above the given resolutions; trainer.TrainStep runs the stages
    at different points of a training step): bit-identical image for the same classes.  Generator.bwd_hooks fire once, largest first."""
    W = _biggan().to(dev).eval()
    G = W.G
    z = GI.rt(551, 3, 120).to(dev)
    cls = torch.tensor([239, 100, 7], device=dev)
    with torch.no_grad():
        ref = W(z, classes=cls)
        h = W.begin(z, pause_res=pauses, classes=cls)
        n, img = 0, None
        while img is None:
            img = W.advance(h)
            n += 1
    assert torch.equal(img, ref)
    assert n == {32: 1, (16, 64): 2, 512: 1}[pauses], n
    sh = (GI.rt(552, 3, 120) * 0.1).to(dev)
    wgt = GI.rt(553, 3, 3, 128, 128).to(dev)
    grads, fired = [], []
    for hooks in (None, [(8, lambda: fired.append(8)), (64, lambda: fired.append(64)), (1024, lambda: fired.append(1024))]):
        s = sh.clone().requires_grad_(True)
        G.bwd_hooks = hooks
        (W(z, s, classes=cls) * wgt).sum().backward()
        assert G.bwd_hooks is None
        grads.append(s.grad.clone())
    assert fired == [1024, 64, 8]
    assert rel_err(grads[1], grads[0]) < 1e-5
