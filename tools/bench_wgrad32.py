"""fp32 weight-gradient kernel: time against the number of K splits (atomics vs parallelism) on the Reconstructor's layer shapes."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0'); B = 32
for ci, co, h in [(64, 64, 64), (128, 128, 32), (256, 256, 16), (512, 512, 8)]:
    x = torch.randn(B, h, h, ci, device=dev); dy = torch.randn(B, h, h, co, device=dev)
    dw = torch.zeros(co, 9, ci, device=dev)
    gf = 2.0 * B * h * h * co * ci * 9 / 1e9
    out = []
    for ks in (0, 4, 8, 16, 32, 64, 128):
        fn = lambda: C.conv2d_wgrad(x, dy, dw, 3, stride=1, pad=1, ksplit=ks, precision=0)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        out.append('ks=%d: %.0f us %.0f TF' % (ks, ms * 1e3, gf / ms))
    print('%d->%d @%d: ' % (ci, co, h) + ' | '.join(out), flush=True)
