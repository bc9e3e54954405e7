"""Micro-benchmark of wgs_sg2_act_bwd on the three largest StyleGAN2-256 layers (B=32)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import _lib as L
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
lib, st = L.lib(), L.stream()
B = 32
for H, C, rgb in [(256, 128, True), (256, 128, False), (128, 256, True), (64, 512, False)]:
    P = H * H
    out = torch.randn(B, P, C, device=dev); gA = torch.randn(B, P, C, device=dev); sA = torch.randn(B, C, device=dev)
    drgb = torch.randn(B, 3, P, device=dev) if rgb else None; wR = torch.randn(3, C, device=dev); sR = torch.randn(B, C, device=dev)
    noise, nw, bias = torch.randn(P, device=dev), torch.ones(1, device=dev), torch.zeros(C, device=dev)
    dy = torch.empty_like(out); num = torch.zeros(B, C, device=dev); dsA = torch.zeros(B, C, device=dev); dsR = torch.zeros(B, C, device=dev)
    dem = torch.rand(B, C, device=dev); am = torch.zeros(1, device=dev)
    ms = timeit(lambda: L.check(lib.wgs_sg2_act_bwd(L.ptr(out), L.ptr(gA), L.ptr(sA), L.ptr(drgb), L.ptr(wR) if rgb else None, L.ptr(sR) if rgb else None,
                                                    L.c_float(1.0), L.ptr(noise), L.ptr(nw), L.ptr(bias), L.ptr(dy), L.ptr(num), L.ptr(dsA),
                                                    L.ptr(dsR) if rgb else None, L.ptr(dem), L.ptr(am), B, P, C, 0, st), 'a'))
    by = 3 * out.numel() * 4
    print('act_bwd %dx%d C=%d rgb=%d: %.3f ms  %.2f TB/s' % (H, H, C, rgb, ms, by / ms / 1e9))
