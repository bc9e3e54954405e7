"""sg2_act_bwd on StyleGAN2-256's layer shapes (B = 32): us and GB/s of its algorithmic traffic (out + gA read, dy written as fp32 or as an fp16 plane)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import _lib as L
dev = torch.device('cuda:0'); lib = L.lib(); B = 32
tot = 0.0
for C_, H, rgb, plane in ((128, 256, True, True), (128, 256, False, False), (256, 128, True, True), (256, 128, False, False), (512, 64, True, False), (512, 64, False, False), (512, 32, True, False), (512, 32, False, False)):
    P = H * H
    out, gA = torch.randn(B, P, C_, device=dev), torch.randn(B, P, C_, device=dev)
    sA, sR = torch.randn(B, C_, device=dev), torch.randn(B, C_, device=dev)
    drgb, wR = torch.randn(B, 3, P, device=dev), torch.randn(3, C_, device=dev)
    noise, nw, bias = torch.randn(P, device=dev), torch.ones(1, device=dev), torch.zeros(C_, device=dev)
    dm = torch.rand(B, C_, device=dev)
    num, dsA, dsR, am = torch.zeros(B, C_, device=dev), torch.zeros(B, C_, device=dev), torch.zeros(B, C_, device=dev), torch.zeros(1, device=dev)
    bound = torch.full((1,), 64.0, device=dev)
    dy = torch.empty(B, P, C_, device=dev, dtype=torch.int16 if plane else torch.float32)
    rgb_args = (L.ptr(drgb if rgb else None), L.ptr(wR) if rgb else None, L.rawptr(sR), L.c_float(0.1 if rgb else 0.0))
    if plane:
        fn = lambda: L.check(lib.wgs_sg2_act_bwd_f16(L.ptr(out), L.ptr(gA), L.rawptr(sA), *rgb_args, L.ptr(noise), L.ptr(nw), L.ptr(bias), L.ptr(dy, torch.int16),
                                                     L.rawptr(bound), L.ptr(num), L.ptr(dsA), L.ptr(dsR), L.ptr(dm), None, B, P, C_, C_, L.stream()), 'act_bwd_f16')
    else:
        fn = lambda: L.check(lib.wgs_sg2_act_bwd(L.ptr(out), L.ptr(gA), L.rawptr(sA), *rgb_args, L.ptr(noise), L.ptr(nw), L.ptr(bias), L.ptr(dy), L.ptr(num),
                                                 L.ptr(dsA), L.ptr(dsR), L.ptr(dm), L.rawptr(am), B, P, C_, C_, L.stream()), 'act_bwd')
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    by = out.numel() * (8 + (2 if plane else 4))
    tot += ms
    print('%3d ch @%3d^2 rgb=%d plane=%d: %6.1f us  %5.2f TB/s' % (C_, H, rgb, plane, ms * 1e3, by / ms / 1e9), flush=True)
print('sum %.3f ms' % tot)
