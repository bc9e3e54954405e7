"""GPU: HIP LeNet reconstructor (train-mode BN, max-pools, two heads) vs the reference golden vectors."""
import pytest
import torch

from tests.test_oracle_golden import _lenet
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('tag', ['cfg1', 'rgb'])
def test_lenet_vs_reference_golden(dev, golden, tag):
    g = golden('reconstructor')
    R, K, B, x1, x2, pl, pm = _lenet(tag)
    R = R.to(dev).train()
    x2d = x2.to(dev).requires_grad_(True)
    logits, mag = R(x1.to(dev), x2d)
    assert logits.shape == (B, K) and mag.shape == (B,)
    ((logits * pl.to(dev)).sum() + (mag * pm.to(dev)).sum()).backward()
    assert rel_err(logits, g['lenet_%s_logits' % tag]) < 1e-4 and rel_err(mag, g['lenet_%s_mag' % tag]) < 1e-4
    assert torch.equal(torch.argmax(logits, 1).cpu(), torch.from_numpy(g['lenet_%s_logits' % tag]).argmax(1))
    dx2 = x2d.grad if tag == 'cfg1' else x2d.grad[:, :, ::4, ::4]
    worst = rel_err(dx2, g['lenet_%s_dx2' % tag])
    scale = max(float(abs(g['lenet_%s_grad_%s' % (tag, n)]).max()) for n, _ in R.named_parameters())
    for n, p in R.named_parameters():
        ref = torch.from_numpy(g['lenet_%s_grad_%s' % (tag, n)])
        pg = p.grad if p.grad.numel() <= 4096 else p.grad.reshape(-1)[::7]
        # a bias in front of a train-mode BatchNorm has an exactly-zero gradient (only round-off noise ~1e-7 on
        # both sides): measure against max(|ref|, 1e-2 * the largest gradient entry of the net)
        e = float((pg.cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-2 * scale)
        worst = max(worst, e)
    print('LeNet %s worst gradient rel err vs reference: %.3e' % (tag, worst))
    assert worst < 2e-3
    for n, b in R.named_buffers():
        if 'running' in n:
            assert rel_err(b, g['lenet_%s_buf_%s' % (tag, n)]) < 1e-4, n
