#!/usr/bin/env python
"""Distribution of the per-sample image error of an arithmetic mode against exact fp32 (max-norm relative, as the 1e-3 gate of
BASELINE.json's north_star is applied in tests/test_precision_schemes_gpu.py), over many latent codes and several random
weight fills.  usage: python tools/err_dist.py [mode=mixed] [n_z=256] [seeds=3]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import golden_inputs as GI
from warpedganspace_amd import conv as C
from warpedganspace_amd.gan_load import StyleGAN2Wrapper
from warpedganspace_amd.stylegan2 import Generator

dev = torch.device('cuda:0')
mode = sys.argv[1] if len(sys.argv) > 1 else 'mixed'
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 256
seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
size = int(os.environ.get('SIZE', 256))
allerr, berr = [], []
for sidx in range(seeds):
    torch.manual_seed(100 + sidx)
    G0 = Generator(size, 512, 8)
    sd = GI.fill_state_dict(G0.state_dict(), 7000 + 13 * sidx)
    for k in sd:
        if k.startswith('style.') and k.endswith('weight'):
            sd[k] = sd[k] * 100.0          # a well-conditioned random mapping network (w = O(1)), as in the tests
    G0.load_state_dict(sd)
    G = StyleGAN2Wrapper(G0.to(dev), False)
    errs = []
    for i in range(0, nz, 32):
        z = torch.randn(min(32, nz - i), 512, device=dev)
        with torch.no_grad():
            ref = G(z, precision='fp32')
            img = G(z, precision=mode)
        errs.append(((img - ref).abs().flatten(1).max(1).values / ref.abs().flatten(1).max(1).values).cpu())
        berr.append(float((img - ref).abs().max() / ref.abs().max()))       # whole-tensor max-norm of the batch (tests.util.rel_err)
    e = torch.cat(errs)
    allerr.append(e)
    print('weights %d: n=%d median %.2e p90 %.2e p99 %.2e max %.2e  over-gate %d' % (
        sidx, e.numel(), float(e.median()), float(e.quantile(0.9)), float(e.quantile(0.99)), float(e.max()), int((e > 1e-3).sum())), flush=True)
e = torch.cat(allerr)
b = torch.tensor(berr)
print('%s @%d per batch of 32 (whole-tensor max-norm, the parity tests\' metric): n=%d median %.2e max %.2e' % (mode, size, b.numel(), float(b.median()), float(b.max())))
print('%s @%d all: n=%d median %.2e p90 %.2e p99 %.2e max %.2e  over-gate %d' % (
    mode, size, e.numel(), float(e.median()), float(e.quantile(0.9)), float(e.quantile(0.99)), float(e.max()), int((e > 1e-3).sum())))
