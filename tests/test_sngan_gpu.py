"""GPU: HIP SNGAN generator (MNIST 32x32 and AnimeFaces 64x64 configurations) vs reference golden / oracle."""
import pytest
import torch
import torch.nn.functional as F

from oracle import wgs_oracle as O
from tests import golden_inputs as GI
from tests.test_oracle_golden import _sngan
from tests.util import rel_err
from warpedganspace_amd.sngan import SNGANWrapper

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('tag', ['mnist', 'anime'])
def test_sngan_vs_reference_golden_and_shared_gate_oracle(dev, golden, tag):
    g = golden('generators')
    G, sd, channels, size, seed = _sngan(tag)
    wrap = SNGANWrapper(G).to(dev).eval()
    assert wrap.dim_z == 128
    z = GI.rt(seed + 1, 3, 128)
    shd = (GI.rt(seed + 2, 3, 128) * 0.1).to(dev).requires_grad_(True)
    G.model.debug_keep = {}
    img = wrap(z.to(dev), shd)
    probe = GI.rt(seed + 3, *img.shape)
    (img * probe.to(dev)).sum().backward()
    ref = g['sngan_%s_img' % tag]
    assert rel_err(img.detach() if size == 32 else F.avg_pool2d(img.detach(), 4), ref) < 1e-4
    e = rel_err(shd.grad, g['sngan_%s_dshift' % tag])
    print('SNGAN %s d/dshift vs reference fp32: %.3e' % (tag, e))
    assert e < 5e-3
    # exact: oracle in float64 through the same ReLU gates
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    sh = (GI.rt(seed + 2, 3, 128) * 0.1).double().requires_grad_(True)
    O.GATE_OVERRIDE = iter([m.cpu() for m in G.model.debug_keep['gates']])
    img_o = O.sngan_generate(sd64, z.double(), sh, channels=channels)
    O.GATE_OVERRIDE = None
    (img_o * probe.double()).sum().backward()
    # pre-tanh activations reach O(50) through three residual up-blocks: fp32 round-off there is ~1e-4 absolute
    assert rel_err(img, img_o.detach()) < 5e-4
    e2 = rel_err(shd.grad, sh.grad)
    print('SNGAN %s shared-gate d/dshift vs fp64 oracle: %.3e' % (tag, e2))
    assert e2 < 1e-3
