"""Winograd F(2x2,3x3) fp32 kernel vs the direct exact-fp32 kernel on the styled 3x3 stride-1 layers of StyleGAN2-256 (B = 32):
time per launch and TFLOP/s counted on the DIRECT form's multiplies (2 * pixels * Cout * Cin * 9) for both.
usage: python tools/bench_wino.py [B]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [(512, 512, 16), (512, 512, 32), (512, 512, 64), (256, 256, 128), (128, 128, 256), (64, 64, 64), (128, 128, 32), (256, 256, 16)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for ci, co, h in SHAPES:
    x = torch.randn(B, h, h, ci, device=dev)
    w = C.pack_weight(torch.randn(co, ci, 3, 3, device=dev) / (9 * ci) ** 0.5)
    cache = C.SplitCache(w)
    s, dm = torch.randn(B, ci, device=dev), torch.rand(B, co, device=dev)
    nz, nw, bias = torch.randn(h * h, device=dev), torch.ones(1, device=dev), torch.zeros(co, device=dev)
    y = torch.empty(B, h, h, co, device=dev)
    epi = dict(a_scale=s, col_scale=dm, bias=bias, noise=nz, noise_w=nw, act_slope=0.2, gain=1.41, out=y, w_split=cache)
    gf = 2.0 * B * h * h * co * ci * 9 / 1e9
    td = timeit(lambda: C.conv2d(x, w, 3, pad=1, precision=0, **epi))
    yd = y.clone()
    tw = timeit(lambda: C.conv2d(x, w, 3, pad=1, precision=C.FP32W, **epi))
    err = float((y - yd).abs().max() / yd.abs().max())
    print('%4d->%4d @%3dx%-3d B%d  direct %8.1f us %6.1f TF | winograd %8.1f us %6.1f TF (%.2fx, executed %.1f TF) | max diff %.1e' % (
        ci, co, h, h, B, td * 1e3, gf / td, tw * 1e3, gf / tw, td / tw, gf / 2.25 / tw, err), flush=True)
