#!/usr/bin/env python
"""Micro-benchmark of the weight-gradient kernels on the ResNet-18 reconstructor's conv shapes at 256x256 input, B=32."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0')
B = 32


def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

for ci, co, h, k, s in [(64, 64, 64, 3, 1), (64, 128, 64, 3, 2), (128, 128, 32, 3, 1), (128, 256, 32, 3, 2), (256, 256, 16, 3, 1),
                        (256, 512, 16, 3, 2), (512, 512, 8, 3, 1), (64, 128, 64, 1, 2)]:
    x = torch.randn(B, h, h, ci, device=dev)
    ho = (h + 2 * (k // 2) - k) // s + 1
    dy = torch.randn(B, ho, ho, co, device=dev)
    dw = torch.zeros(co, k * k, ci, device=dev)
    fl = 2.0 * B * ho * ho * co * ci * k * k
    line = 'wgrad %3d->%3d @%2d k%d s%d: ' % (ci, co, h, k, s)
    for prec in (0, 1):
        ms = timeit(lambda: C.conv2d_wgrad(x, dy, dw, k, stride=s, pad=k // 2, precision=prec))
        line += ' %s %7.3f ms %6.1f TF |' % ('fp32  ' if prec == 0 else 'bf16x3', ms, fl / ms / 1e9)
    if prec == 1:      # agreement of the two kernels on this shape
        d0 = torch.zeros_like(dw); d1 = torch.zeros_like(dw)
        C.conv2d_wgrad(x, dy, d0, k, stride=s, pad=k // 2, precision=0); C.conv2d_wgrad(x, dy, d1, k, stride=s, pad=k // 2, precision=1)
        line += ' diff %.1e' % ((d1 - d0).abs().max() / d0.abs().max()).item()
    print(line, flush=True)
