"""Split-bf16 F(2,3) x direct kernel (conv_wino_bf16.hip, 'bf16x3w') vs the direct split-bf16 kernels ('bf16x3') and plain fp16 on the styled
3x3 stride-1 layers of StyleGAN2-256 (B = 32): time per launch and TFLOP/s counted on the DIRECT form's multiplies (2 * pixels * Cout * Cin * 9).
usage: python tools/bench_wino16.py [B]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [(512, 512, 32), (512, 512, 64), (256, 256, 128), (128, 128, 256)]


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


ONLY = bool(os.environ.get('W16ONLY'))       # ablation builds (WGS_LIB=tools/_bin/libwgs_w16abl<n>.so): time the new kernel only
for ci, co, h in SHAPES:
    x = torch.randn(B, h, h, ci, device=dev)
    w = C.pack_weight(torch.randn(co, ci, 3, 3, device=dev) / (9 * ci) ** 0.5)
    cache = C.SplitCache(w)
    s, dm = torch.randn(B, ci, device=dev), torch.rand(B, co, device=dev)
    nz, nw, bias = torch.randn(h * h, device=dev), torch.ones(1, device=dev), torch.zeros(co, device=dev)
    y = torch.empty(B, h, h, co, device=dev)
    amax, smax = x.abs().max().reshape(1), s.abs().max().reshape(1)
    epi = dict(a_scale=s, col_scale=dm, bias=bias, noise=nz, noise_w=nw, act_slope=0.2, gain=1.41, out=y, w_split=cache)
    gf = 2.0 * B * h * h * co * ci * 9 / 1e9
    if ONLY:
        tw = timeit(lambda: C.conv2d(x, w, 3, pad=1, precision=C.BF16W, **epi))
        print('%4d->%4d @%3dx%-3d B%d  bf16x3w %7.1f us' % (ci, co, h, h, B, tw * 1e3), flush=True)
        continue
    t0 = timeit(lambda: C.conv2d(x, w, 3, pad=1, precision=0, **epi))
    y0 = y.clone()
    td = timeit(lambda: C.conv2d(x, w, 3, pad=1, precision=1, **epi))
    ed = float((y - y0).abs().max() / y0.abs().max())
    tw = timeit(lambda: C.conv2d(x, w, 3, pad=1, precision=C.BF16W, **epi))
    ew = float((y - y0).abs().max() / y0.abs().max())
    tf = timeit(lambda: C.conv2d(x, w, 3, pad=1, precision=2, a_amax=amax, a_amax2=smax, **epi))
    print('%4d->%4d @%3dx%-3d B%d  bf16x3 %7.1f us %6.1f TF (err %.1e) | bf16x3w %7.1f us %6.1f TF (%.2fx; executed %.0f TF of 2500; err %.1e) | f16 %7.1f us | fp32 %7.1f us' % (
        ci, co, h, h, B, td * 1e3, gf / td, ed, tw * 1e3, gf / tw, td / tw, gf * 2 / tw, ew, tf * 1e3, t0 * 1e3), flush=True)
