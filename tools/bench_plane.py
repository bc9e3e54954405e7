#!/usr/bin/env python
"""Micro-benchmark: stride-1 3x3 conv in plain fp16 fed (a) with the fp32 activation tensor + style vector (patch kernel: style
multiply, scale, conversion while staging), (b) with a producer-written fp16 plane through the patch kernel (XF16 form: staged as it
is), (c) with the plane through the LDS-DMA kernel.  B = 32, StyleGAN2-256's layer shapes.  usage: python tools/bench_plane.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C

dev = torch.device('cuda:0')
B = int(os.environ.get('B', 32))


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def flags(**env):
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    L.lib().wgs_dev_reload_flags()


am = torch.ones(1, device=dev) * 4.0
ONLY = int(os.environ.get('ONLY', 0))      # ONLY=128: the 128 -> 128 @256^2 layer alone
for ci, co, h in [(512, 512, 64), (256, 256, 128), (128, 128, 256)]:
    if ONLY and ci != ONLY:
        continue
    x = torch.randn(B, h, h, ci, device=dev)
    w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    s = torch.randn(B, ci, device=dev); dm = torch.rand(B, co, device=dev)
    y = torch.empty(B, h, h, co, device=dev)
    nz, nw, bias = torch.randn(h * h, device=dev), torch.ones(1, device=dev), torch.zeros(co, device=dev)
    ws = C.split_weight(w, 2)
    plane = (x * 512.0).half().view(torch.int16)
    fl = 2.0 * B * h * h * co * ci * 9
    epi = dict(col_scale=dm, noise=nz, noise_w=nw, bias=bias, act_slope=0.2, gain=1.41, precision=2, w_split=ws, out=y)
    L.lib().wgs_dev_trace_kernels(1)
    res = []
    t = timeit(lambda: C.conv2d(x, w, 3, pad=1, a_scale=s, a_amax=am, **epi)); res.append(('fp32 x + style', t, L.lib().wgs_dev_last_kernel().decode()))
    flags(WGS_PLANE_PATCH_MAX_CO=100000)
    t = timeit(lambda: C.conv2d(plane, w, 3, pad=1, a_amax=am, x_f16=True, **epi)); res.append(('fp16 plane, patch', t, L.lib().wgs_dev_last_kernel().decode()))
    flags(WGS_PLANE_PATCH_MAX_CO=0)
    t = timeit(lambda: C.conv2d(plane, w, 3, pad=1, a_amax=am, x_f16=True, **epi)); res.append(('fp16 plane, LDS-DMA', t, L.lib().wgs_dev_last_kernel().decode()))
    flags(WGS_PLANE_PATCH_MAX_CO=None)
    L.lib().wgs_dev_trace_kernels(0)
    print('conv3x3 f16 %4d->%4d @%3d:' % (ci, co, h))
    for name, t, k in res:
        print('    %-22s %7.3f ms %7.1f TF   %s' % (name, t, fl / t / 1e9, k), flush=True)
