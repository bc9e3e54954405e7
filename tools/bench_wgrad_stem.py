"""The stem's weight gradient (7x7/2, 8 -> 64 channels, s2d input) in exact fp32 and split-bf16: us per launch at 256^2 / 1024^2 inputs."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0')
for B, S in ((32, 256), (8, 1024), (32, 1024)):
    xs = torch.randn(B, S // 2, S // 2, 32, device=dev)
    gy = torch.randn(B, S // 2, S // 2, 64, device=dev)
    for prec in (0, 1):
        dw = torch.zeros(64, 49, 8, device=dev)
        for _ in range(3):
            C.conv2d_wgrad(xs, gy, dw, 7, stride=2, pad=3, x_s2d=True, precision=prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            C.conv2d_wgrad(xs, gy, dw, 7, stride=2, pad=3, x_s2d=True, precision=prec)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        fl = 2.0 * B * (S // 2) ** 2 * 64 * 8 * 49
        print('B %d %dx%d 7x7 form precision %d: %.0f us, %.1f TFLOP/s' % (B, S, S, prec, us, fl / us / 1e6))
        dws = torch.zeros(64, 16, 32, device=dev)
        for _ in range(3):
            C.conv2d_wgrad(xs, gy, dws, 4, stride=1, pad=2, precision=prec)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            C.conv2d_wgrad(xs, gy, dws, 4, stride=1, pad=2, precision=prec)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print('B %d %dx%d s2d form precision %d: %.0f us, %.1f TFLOP/s of the 7x7 form' % (B, S, S, prec, us, fl / us / 1e6))
