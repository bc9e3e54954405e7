"""Weight gradients of the Reconstructor's stride-1 3x3 convs: the direct-fragment kernel (conv_wgrad_direct.hip) against the LDS-staged
kernels it replaces (conv_wgrad16.hip split-bf16, conv_igemm.hip exact fp32), at 256^2 and 1024^2 generator outputs (B = 32 / 8).
usage: python tools/bench_wgrad_direct.py [ksweep]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0')
lib = L.lib()


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def staged(on):
    if on:
        os.environ['WGS_WGRAD_STAGED'] = '1'
    else:
        os.environ.pop('WGS_WGRAD_STAGED', None)
    lib.wgs_dev_reload_flags()


tot = {}
for B, scale in ((32, 1), (8, 4)):
    for ci, h, n in ((64, 64, 4), (128, 32, 3), (256, 16, 3), (512, 8, 3)):
        h *= scale
        x = torch.randn(B, h, h, ci, device=dev); dy = torch.randn(B, h, h, ci, device=dev)
        dw = torch.zeros(ci, 9, ci, device=dev)
        gf = 2.0 * B * h * h * ci * ci * 9 / 1e9
        row = []
        for prec in (1, 0):
            for st in (False, True):
                staged(st)
                ms = timed(lambda: C.conv2d_wgrad(x, dy, dw, 3, stride=1, pad=1, precision=prec))
                row.append((ms, gf / ms))
                tot[(scale, prec, st)] = tot.get((scale, prec, st), 0.0) + n * ms
        staged(False)
        print('B=%d %d->%d @%d (x%d per step): split-bf16 direct %.0f us %.0f TF | staged %.0f us %.0f TF || fp32 direct %.0f us %.0f TF | staged %.0f us %.0f TF' % (
            B, ci, ci, h, n, row[0][0] * 1e3, row[0][1], row[1][0] * 1e3, row[1][1], row[2][0] * 1e3, row[2][1], row[3][0] * 1e3, row[3][1]), flush=True)
        if len(sys.argv) > 1 and sys.argv[1] == 'ksweep':
            for prec in (1, 0):
                out = []
                for ks in (1, 2, 4, 8, 16, 32, 64, 128, 256):
                    ms = timed(lambda: C.conv2d_wgrad(x, dy, dw, 3, stride=1, pad=1, precision=prec, ksplit=ks), n=5)
                    out.append('%d:%.0f' % (ks, ms * 1e3))
                print('    ksplit sweep prec %d (us): %s' % (prec, ' '.join(out)), flush=True)
for k in sorted(tot):
    print('sum over the 13 stride-1 layers, inputs x%d, precision %d, %s: %.3f ms per step' % (k[0], k[1], 'staged' if k[2] else 'direct', tot[k]))
