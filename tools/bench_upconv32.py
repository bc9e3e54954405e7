"""fp32 up-conv phases (merged launch) A/B: time of conv_transpose2d_s2 on the StyleGAN2-256 up-sampling shapes."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0'); B = 32
for ci, co, h in [(512, 512, 32), (512, 256, 64), (256, 128, 128)]:
    x = torch.randn(B, h, h, ci, device=dev); w = C.pack_weight(torch.randn(co, ci, 3, 3, device=dev) / (9 * ci) ** 0.5)
    s, dm = torch.randn(B, ci, device=dev), torch.rand(B, co, device=dev)
    t = torch.empty(B, 2 * h + 1, 2 * h + 1, co, device=dev)
    fn = lambda: C.conv_transpose2d_s2(x, w, out=t, a_scale=s, col_scale=dm, precision=0)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    gf = 2.0 * B * h * h * 9 * ci * co / 1e9
    print('%d->%d @%d up-conv phases: %.1f us %.1f TF  checksum %.6e' % (ci, co, h, ms * 1e3, gf / ms, float(t.double().abs().sum())), flush=True)
