#!/usr/bin/env python
"""Micro-benchmark of the 16-bit-operand conv kernels on the StyleGAN2-256 layer shapes (B=32): stride-1 3x3 (patch form),
transposed conv (merged phases) and its stride-2 dgrad, per arithmetic mode.  usage: python tools/bench_conv16.py [modes]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C

dev = torch.device('cuda:0')
B = int(os.environ.get('B', 32))
modes = [C.precision_code(m) for m in (sys.argv[1:] or ['bf16x3', 'f16'])]


def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

amax = torch.ones(1, device=dev) * 4.0
for ci, co, h in [(512, 512, 16), (512, 512, 32), (512, 512, 64), (256, 256, 128), (128, 128, 256)]:
    x = torch.randn(B, h, h, ci, device=dev)
    w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    s = torch.randn(B, ci, device=dev); dm = torch.rand(B, co, device=dev)
    y = torch.empty(B, h, h, co, device=dev)
    nz, nw, bias = torch.randn(h * h, device=dev), torch.ones(1, device=dev), torch.zeros(co, device=dev)
    fl = 2.0 * B * h * h * co * ci * 9
    line = 'conv3x3 %4d->%4d @%3d: ' % (ci, co, h)
    for m in modes:
        ws = C.split_weight(w, m)
        ms = timeit(lambda: C.conv2d(x, w, 3, pad=1, out=y, a_scale=s, col_scale=dm, noise=nz, noise_w=nw, bias=bias, act_slope=0.2, gain=1.41, precision=m, w_split=ws))
        line += ' %s %7.3f ms %6.1f TF |' % (C.precision_name(m), ms, fl / ms / 1e9)
    print(line, flush=True)
for ci, co, h in [(512, 512, 32), (512, 256, 64), (256, 128, 128)]:
    x = torch.randn(B, h, h, ci, device=dev)
    w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    s = torch.randn(B, ci, device=dev); dm = torch.rand(B, co, device=dev)
    fl = 2.0 * B * h * h * co * ci * 9
    line = 'convT s2 %4d->%4d @%3d->%3d: ' % (ci, co, h, 2 * h + 1)
    t = torch.empty(B, 2 * h + 1, 2 * h + 1, co, device=dev)
    for m in modes:
        ws = C.split_weight(w, m)
        ms = timeit(lambda: C.conv_transpose2d_s2(x, w, out=t, a_scale=s, col_scale=dm, precision=m, w_split=ws))
        line += ' %s %7.3f ms %6.1f TF |' % (C.precision_name(m), ms, fl / ms / 1e9)
    print(line, flush=True)
    wt = C.repack_w_t(w, co, 9, ci)
    line = '   its stride-2 dgrad: '
    for m in modes:
        wts = C.split_weight(wt, m)
        ms = timeit(lambda: C.conv_transpose2d_s2_dgrad(t, wt, precision=m, w_split=wts, a_amax=amax, a_bound=4.0))
        line += ' %s %7.3f ms %6.1f TF |' % (C.precision_name(m), ms, fl / ms / 1e9)
    print(line, flush=True)
