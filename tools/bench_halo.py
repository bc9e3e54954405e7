#!/usr/bin/env python
"""Micro-benchmark: the few-channel 3x3 stride-1 kernel (conv_halo16.hip) against the GEMM-tiled route on the StyleGAN2-1024 (B = 8)
and ProgGAN-1024 (B = 32) layer shapes; HBM floor = (input + output tensor) / 6.3 TB/s.  usage: python tools/bench_halo.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0')


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for B, ci, co, h, styled in [(8, 64, 64, 512, True), (8, 32, 32, 1024, True), (32, 64, 64, 256, False), (32, 32, 32, 512, False), (8, 16, 16, 1024, False)]:
    x = torch.randn(B, h, h, ci, device=dev)
    w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    s = torch.randn(B, ci, device=dev) if styled else None
    dm = torch.rand(B, co, device=dev) if styled else None
    y = torch.empty(B, h, h, co, device=dev)
    nz, nw, bias = torch.randn(h * h, device=dev), torch.ones(1, device=dev), torch.zeros(co, device=dev)
    am = x.abs().amax().reshape(1)
    fl = 2.0 * B * h * h * co * ci * 9
    floor = (x.numel() + y.numel()) * 4 / 6.3e12 * 1e3
    print('conv3x3 %2d->%2d @%4d B%d%s: %.1f GFLOP, HBM floor %.3f ms' % (ci, co, h, B, ' styled' if styled else '', fl / 1e9, floor))
    for m in (1, 2, 3):
        ws = C.split_weight(w, m)
        res = []
        for halo in (True, False):
            if halo: os.environ.pop('WGS_NO_HALO', None)
            else: os.environ['WGS_NO_HALO'] = '1'
            L.lib().wgs_dev_reload_flags()
            L.lib().wgs_dev_trace_kernels(1)
            t = timeit(lambda: C.conv2d(x, w, 3, pad=1, out=y, a_scale=s, col_scale=dm, noise=nz if styled else None, noise_w=nw if styled else None, bias=bias,
                                        act_slope=0.2, gain=1.41, precision=m, w_split=ws, a_amax=am))
            res.append((t, L.lib().wgs_dev_last_kernel().decode()))
            L.lib().wgs_dev_trace_kernels(0)
        os.environ.pop('WGS_NO_HALO', None); L.lib().wgs_dev_reload_flags()
        print('   %-7s halo %.3f ms (%.0f TF, %.2fx floor)  | GEMM-tiled %.3f ms (%.0f TF)   %s | %s' % (
            C.precision_name(m), res[0][0], fl / res[0][0] / 1e9, res[0][0] / floor, res[1][0], fl / res[1][0] / 1e9, res[0][1], res[1][1]), flush=True)
