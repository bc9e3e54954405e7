#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM conv kernel on the StyleGAN2-256 layer shapes (B=32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C

dev = torch.device('cuda:0')
B = int(os.environ.get('B', 32))
shapes = [(512, 512, 4), (512, 512, 8), (512, 512, 16), (512, 512, 32), (512, 512, 64), (256, 256, 128), (128, 128, 256),
          (64, 64, 64), (128, 128, 32)]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

for ci, co, h in shapes:
    x = torch.randn(B, h, h, ci, device=dev)
    w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    s = torch.randn(B, ci, device=dev)
    y = torch.empty(B, h, h, co, device=dev)
    ms = timeit(lambda: C.conv2d(x, w, 3, pad=1, out=y))
    ms2 = timeit(lambda: C.conv2d(x, w, 3, pad=1, out=y, a_scale=s, col_scale=s[:, :co].contiguous(), act_slope=0.2, gain=1.41))
    ms3 = timeit(lambda: C.conv2d(x, w, 3, pad=1, out=y, a_scale=s, col_scale=s[:, :co].contiguous(), act_slope=0.2, gain=1.41, precision=1))
    fl = 2.0 * B * h * h * co * ci * 9
    print('conv %4d->%4d @%3d  M=%8d  %8.3f ms  %6.1f TF | fused %8.3f ms %6.1f TF' % (ci, co, h, B * h * h, ms, fl / ms / 1e9, ms2, fl / ms2 / 1e9) + ' | split-bf16 %8.3f ms %6.1f TF' % (ms3, fl / ms3 / 1e9))
    if h <= 64:
        xt = torch.randn(B, h, h, ci, device=dev)
        ms = timeit(lambda: C.conv_transpose2d_s2(xt, w))
        print('   convT s2 %4d->%4d @%3d->%3d  %8.3f ms  %6.1f TF' % (ci, co, h, 2 * h + 1, ms, fl / ms / 1e9))
    dy = torch.randn(B, h, h, co, device=dev)
    dw = torch.zeros(co, 9, ci, device=dev)
    ms = timeit(lambda: C.conv2d_wgrad(x, dy, dw, 3, pad=1))
    print('   wgrad %8.3f ms  %6.1f TF' % (ms, fl / ms / 1e9))
