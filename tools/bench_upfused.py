#!/usr/bin/env python
"""Micro-benchmark: StyleGAN2 up-sampling layer, fused kernel (conv_upfused.hip) vs phase GEMMs + blur kernel, B=32.
usage: python tools/bench_upfused.py [f16 f16x2]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C, _lib as L

dev = torch.device('cuda:0')
B = int(os.environ.get('B', 32))
modes = [C.precision_code(m) for m in (sys.argv[1:] or ['f16', 'f16x2'])]


def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

k1 = torch.tensor([1., 3., 3., 1.])
kern = (k1[:, None] * k1[None, :] / 64 * 4).to(dev)
for ci, co, h in [(512, 512, 16), (512, 512, 32), (512, 256, 64), (256, 128, 128)]:
    x = torch.randn(B, h, h, ci, device=dev)
    w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    s = torch.randn(B, ci, device=dev); dm = torch.rand(B, co, device=dev)
    nz, nw, bias = torch.randn(4 * h * h, device=dev), torch.ones(1, device=dev), torch.zeros(co, device=dev)
    fl = 2.0 * B * h * h * co * ci * 9
    t = torch.empty(B, 2 * h + 1, 2 * h + 1, co, device=dev)
    y = torch.empty(B, 2 * h, 2 * h, co, device=dev)
    am = torch.zeros(1, device=dev)
    line = 'up %4d->%4d @%3d->%3d: ' % (ci, co, h, 2 * h)
    for m in modes:
        ws = C.split_weight(w, m)

        def unfused():
            C.conv_transpose2d_s2(x, w, out=t, a_scale=s, a_ld=ci, col_scale=dm, precision=m, w_split=ws)
            L.check(L.lib().wgs_sg2_blur_noise_bias_act(L.ptr(t), L.ptr(kern), L.ptr(nz), L.ptr(nw), L.ptr(bias), L.ptr(y), L.ptr(am),
                                                        B, 2 * h, 2 * h, co, L.stream()), 'blur')
        ms_u = timeit(unfused)
        y_u = y.clone()
        ms_f = timeit(lambda: C.upconv_blur_act(x, ws, kern, s, ci, dm, nz, nw, bias, m, y_amax=am))
        y_f = C.upconv_blur_act(x, ws, kern, s, ci, dm, nz, nw, bias, m)
        err = ((y_f - y_u).abs().max() / y_u.abs().max()).item()
        line += ' %s unfused %6.3f ms | fused %6.3f ms %6.1f TF (diff %.1e) |' % (C.precision_name(m), ms_u, ms_f, fl / ms_f / 1e9, err)
    print(line, flush=True)
