#!/usr/bin/env python
"""Candidate per-layer arithmetic policies of a StyleGAN2 generator ('mixed', conv.MixedPolicy): image error against the exact-fp32
kernels over many latent codes (max-norm relative; per batch tensor as the parity tests apply the 1e-3 gate, and per single image)
and the time of the full training step under each policy.
usage: python tools/policy_sweep.py SIZE NZ BATCH [name=res:s1,up;res:s1,up ...] ...   (codes: 1 bf16x3, 2 f16, 3 f16x2, w = 7 bf16x3w; no table = uniform mode name;
       'ladder' = every rung of conv.STRICT_LADDER[SIZE]).  INIT=bench: bench.py's generator (raw constructor initialisation under torch.manual_seed(0)) instead of
       the well-conditioned fills; BELOW=1: the listed tables with direct split-bf16 below them (round 5's form) instead of the F(2,3) form.
e.g.   python tools/policy_sweep.py 256 192 32 cur=64:2,3;128:2,3;256:2,3 f16x2 a=128:2,3;256:2,3"""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import golden_inputs as GI
from warpedganspace_amd import conv as C
from warpedganspace_amd.gan_load import StyleGAN2Wrapper
from warpedganspace_amd.reconstructor import Reconstructor
from warpedganspace_amd.stylegan2 import Generator
from warpedganspace_amd.support_sets import SupportSets
from warpedganspace_amd.trainer import TrainStep

dev = torch.device('cuda:0')
size, nz, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
seeds = int(os.environ.get('SEEDS', 2))
K, N = (200, 64) if size == 1024 else (128, 32)


def parse(spec):
    if '=' not in spec:
        return spec, spec, None
    name, tab = spec.split('=', 1)
    table = {}
    for e in tab.split(';'):
        r, m = e.split(':')
        s1, up = m.split(',')
        table[int(r)] = tuple(C.BF16W if v == 'w' else int(v) for v in (s1, up))
    return name, 'mixed', C.MixedPolicy(table, bwd_table=({64: (2, 2), 128: (2, 2), 256: (2, 2)} if size <= 256 else {64: (2, 2), 128: (2, 2), 256: (2, 2), 512: (3, 2), 1024: (3, 2)}), **({'below': 1} if os.environ.get('BELOW') == '1' else {}))


cands = []
for a in sys.argv[4:]:
    if a == 'ladder':
        cands += [(n_.split(' ')[-1] if n_.startswith('default') else n_, 'mixed', p_) for n_, p_ in C.STRICT_LADDER[size]]
    else:
        cands.append(parse(a))
cands = cands or [('mixed', 'mixed', None)]
gens = []
if os.environ.get('INIT') == 'bench':
    from warpedganspace_amd.gan_load import build_stylegan2
    torch.manual_seed(0)
    gens.append(build_stylegan2(None, resolution=size).to(dev).eval())
    seeds = 0
for sidx in range(seeds):
    torch.manual_seed(100 + sidx)
    G0 = Generator(size, 512, 8)
    sd = GI.fill_state_dict(G0.state_dict(), 7000 + 13 * sidx)
    for k in sd:
        if k.startswith('style.') and k.endswith('weight'):
            sd[k] = sd[k] * 100.0          # a well-conditioned random mapping network (w = O(1)), as in the tests
    G0.load_state_dict(sd)
    gens.append(StyleGAN2Wrapper(G0.to(dev).eval(), False))
torch.manual_seed(12345)
zs = [torch.randn(min(B, nz - i), 512, device=dev) for i in range(0, nz, B)]
refs = {}
for name, mode, pol in cands:
    per, bat = [], []
    for gi, G in enumerate(gens):
        for zi, z in enumerate(zs):
            with torch.no_grad():
                if (gi, zi) not in refs:
                    refs[(gi, zi)] = G(z, precision='fp32').cpu() if size > 256 else G(z, precision='fp32')
                ref = refs[(gi, zi)].to(dev)
                img = G(z, precision=mode, **({'policy': pol} if pol is not None else {}))
            per.append(((img - ref).abs().flatten(1).max(1).values / ref.abs().flatten(1).max(1).values).cpu())
            bat.append(float((img - ref).abs().max() / ref.abs().max()))
            del img, ref
    e, b = torch.cat(per), torch.tensor(bat)
    # step time under this policy
    G = gens[0]
    G.G.mixed_policy = pol          # (the step engine below runs precision `mode`: an explicit 'mixed' reads the instance's table)
    p = types.SimpleNamespace(reconstructor_lr=1e-4, support_set_lr=1e-4, min_shift_magnitude=0.25, max_shift_magnitude=0.45, lambda_cls=1.0,
                              lambda_reg=0.25, z_truncation=None, shift_in_w_space=False)
    S = SupportSets(K, N, 512, learn_alphas=False, learn_gammas=True, gamma=1.0 / 512)
    R = Reconstructor('ResNet', K)
    eng = TrainStep(G, S.to(dev).train(), R.to(dev).train(), p, B, dev, seed=0, precision=mode)
    for _ in range(4):
        eng.step()
    torch.cuda.synchronize()
    n = 10 if size <= 256 else 5
    t0 = time.perf_counter()
    for _ in range(n):
        eng.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    del eng, S, R
    G.G.mixed_policy = None
    torch.cuda.empty_cache()
    print('%-10s @%d step %.2f ms | batch(B=%d) median %.2e max %.2e | image n=%d median %.2e p90 %.2e p99 %.2e max %.2e over-gate %.1f%%' % (
        name, size, ms, B, float(b.median()), float(b.max()), e.numel(), float(e.median()), float(e.quantile(0.9)), float(e.quantile(0.99)),
        float(e.max()), 100.0 * float((e > 1e-3).float().mean())), flush=True)
