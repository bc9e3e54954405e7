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
