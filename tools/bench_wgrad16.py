"""Split-bf16 weight-gradient kernels (conv_wgrad16.hip) on the Reconstructor's layer shapes at 256^2 and 1024^2 inputs: us and TFLOP/s
per launch.  WGS_WGRAD_SHALLOW=1 selects the one-register-set form of the kernel-row kernel (development A/B)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0'); B = 32
for scale in (1, 4):
    for ci, co, h, k, s in [(64, 64, 64, 3, 1), (64, 128, 64, 3, 2), (128, 128, 32, 3, 1), (128, 256, 32, 3, 2), (256, 256, 16, 3, 1), (512, 512, 8, 3, 1), (64, 128, 64, 1, 2)]:
        h = h * scale
        x = torch.randn(B, h, h, ci, device=dev); ho = h // s
        dy = torch.randn(B, ho, ho, co, device=dev)
        dw = torch.zeros(co, k * k, ci, device=dev)
        gf = 2.0 * B * ho * ho * co * ci * k * k / 1e9
        fn = lambda: C.conv2d_wgrad(x, dy, dw, k, stride=s, pad=k // 2, precision=1)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print('%d->%d @%d k%d s%d: %.0f us %.0f TF' % (ci, co, h, k, s, ms * 1e3, gf / ms), flush=True)
