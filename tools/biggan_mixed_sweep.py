"""BigGAN 'mixed' policy sweep: image error (per image, max-norm relative to exact fp32) and forward time with fp16 x2 in the blocks whose
output is >= R, split-bf16 below.  usage: python tools/biggan_mixed_sweep.py [resolution=128]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd.biggan import build_biggan
dev = torch.device('cuda:0')
res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(1)
if res == 128:
    G = build_biggan(None, (239,)).to(dev).eval()
else:
    from warpedganspace_amd.biggan import BigGANWrapper, Generator
    G = BigGANWrapper(Generator(G_ch=96, dim_z=120, shared_dim=128, hier=True, G_attn='64', BN_eps=1e-5, SN_eps=1e-6, resolution=res, n_classes=1000), (239,)).to(dev).eval()
B = 32 if res == 128 else 16
zs = [torch.randn(B, G.dim_z, device=dev) for _ in range(6)]
with torch.no_grad():
    refs = [G(z, precision='fp32') for z in zs]
    for name, frm in [('bf16x3', None)] + [('mixed', r) for r in (res, res // 2, res // 4, res // 8, 8)]:
        G.G.mixed_from_res = frm
        errs = []
        for z, ref in zip(zs, refs):
            img = G(z, precision=name)
            errs.append(((img - ref).abs().flatten(1).max(1).values / ref.abs().flatten(1).max(1).values).cpu())
        e = torch.cat(errs)
        for _ in range(3): G(zs[0], precision=name)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): G(zs[0], precision=name)
        b.record(); torch.cuda.synchronize()
        print('BigGAN-%d %s from %s: per image median %.2e p99 %.2e max %.2e over-gate %.3f | forward %.2f ms (B=%d)' % (
            res, name, frm, float(e.median()), float(e.quantile(0.99)), float(e.max()), float((e > 1e-3).float().mean()), a.elapsed_time(b) / 10, B), flush=True)
