#!/usr/bin/env python
"""Exact-fp32 conv kernels on the StyleGAN2-256 / ResNet-18 layer shapes (B=32): the plain three-phase kernel (conv_igemm.hip,
WGS_F32_OLD) against the slot-interleaved one (conv_igemm_f32.hip), 4-wave 128-row tiles only (WGS_F32_SMALL) and the default policy (8-wave 256 x 256 tiles where Cout allows, merged up-conv phases);
checks the new results against the old ones.  usage: python tools/bench_conv32.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C

dev = torch.device('cuda:0')
B = int(os.environ.get('B', 32))
FORMS = [('old', {'WGS_F32_OLD': '1'}), ('small', {'WGS_F32_SMALL': '1'}), ('default', {})]


def use(env):
    for k in ('WGS_F32_OLD', 'WGS_F32_SMALL'):
        os.environ.pop(k, None)
    os.environ.update(env)
    L.lib().wgs_dev_reload_flags()


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def row(name, fl, fn, out):
    line, ref = name, None
    for form, env in FORMS:
        use(env)
        ms = timeit(fn)
        y = out().clone()
        if ref is None:
            ref = y
        err = float((y - ref).abs().max() / ref.abs().max())
        line += ' | %-7s %7.3f ms %6.1f TF err %.1e' % (form, ms, fl / ms / 1e9, err)
    print(line, flush=True)


for ci, co, h in [(512, 512, 8), (512, 512, 16), (512, 512, 32), (512, 512, 64), (256, 256, 128), (128, 128, 256)]:
    x = torch.randn(B, h, h, ci, device=dev)
    w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    s = torch.randn(B, ci, device=dev); dm = torch.rand(B, co, device=dev)
    y = torch.empty(B, h, h, co, device=dev)
    nz, nw, bias = torch.randn(h * h, device=dev), torch.ones(1, device=dev), torch.zeros(co, device=dev)
    fl = 2.0 * B * h * h * co * ci * 9
    row('conv3x3 styled %4d->%4d @%3d' % (ci, co, h), fl,
        lambda: C.conv2d(x, w, 3, pad=1, out=y, a_scale=s, col_scale=dm, noise=nz, noise_w=nw, bias=bias, act_slope=0.2, gain=1.41, precision=0), lambda: y)
    wt = C.repack_w_t(w, co, 9, ci)
    holder = {}

    def dgrad():
        holder['y'] = C.conv2d_dgrad(y, wt, (h, h), 3, pad=1, precision=0)
    row('   its dgrad (plain)          ', fl, dgrad, lambda: holder['y'])
for ci, co, h in [(512, 512, 16), (512, 512, 32), (512, 256, 64), (256, 128, 128)]:
    x = torch.randn(B, h, h, ci, device=dev)
    w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    s = torch.randn(B, ci, device=dev); dm = torch.rand(B, co, device=dev)
    fl = 2.0 * B * h * h * co * ci * 9
    t = torch.empty(B, 2 * h + 1, 2 * h + 1, co, device=dev)
    row('convT s2 %4d->%4d @%3d->%3d' % (ci, co, h, 2 * h + 1), fl, lambda: C.conv_transpose2d_s2(x, w, out=t, a_scale=s, col_scale=dm, precision=0), lambda: t)
    wt = C.repack_w_t(w, co, 9, ci)
    holder = {}

    def dg():
        holder['y'] = C.conv_transpose2d_s2_dgrad(t, wt, precision=0)
    row('   its stride-2 dgrad         ', fl, dg, lambda: holder['y'])
# ResNet-18 (R) forward convs at 256^2 inputs, B=32: (ci, co, h_in, k, stride)
for ci, co, h, k, st in [(64, 64, 64, 3, 1), (64, 128, 64, 3, 2), (128, 128, 32, 3, 1), (128, 256, 32, 3, 2), (256, 256, 16, 3, 1), (256, 512, 16, 3, 2),
                         (512, 512, 8, 3, 1), (64, 128, 64, 1, 2)]:
    x = torch.randn(B, h, h, ci, device=dev)
    w = torch.randn(co, k * k, ci, device=dev) / (k * k * ci) ** 0.5
    ho = (h + 2 * (k // 2) - k) // st + 1
    y = torch.empty(B, ho, ho, co, device=dev)
    fl = 2.0 * B * ho * ho * co * ci * k * k
    row('R conv %dx%d/%d %4d->%4d @%3d' % (k, k, st, ci, co, h), fl, lambda: C.conv2d(x, w, k, stride=st, pad=k // 2, out=y, precision=0), lambda: y)
use({})
