#!/usr/bin/env python
"""ToRGB (+ fused skip up-sampling) at the StyleGAN2-256 resolutions, B = 32: time and bytes per launch.  usage: python tools/bench_torgb.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import _lib as L

dev = torch.device('cuda:0')
lib = L.lib()
B = 32
k1 = torch.tensor([1., 3., 3., 1.]); upk = (k1[:, None] * k1[None, :] / 64 * 4).to(dev).contiguous()
for H, Cc in [(8, 512), (16, 512), (32, 512), (64, 512), (128, 256), (256, 128)]:
    x = torch.randn(B, H * H, Cc, device=dev)
    s = torch.randn(B, Cc, device=dev); w = torch.randn(3, Cc, device=dev); bias = torch.zeros(3, device=dev)
    lo = torch.randn(B, 3, H // 2, H // 2, device=dev); img = torch.empty(B, 3, H, H, device=dev)
    fn = lambda: L.check(lib.wgs_sg2_torgb_up_fwd(L.ptr(x), L.ptr(s), Cc, L.ptr(w), L.ptr(bias), L.ptr(lo), L.ptr(upk), L.ptr(img), B, H, H, Cc,
                                                 L.c_float(0.05), L.stream()), 'torgb_up')
    fn(); fn(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(20):
        fn()
    en.record(); torch.cuda.synchronize()
    us = st.elapsed_time(en) / 20 * 1e3
    print('torgb+up %3d^2 C=%3d: %7.1f us  %6.0f GB/s' % (H, Cc, us, x.numel() * 4 / us / 1e3), flush=True)
