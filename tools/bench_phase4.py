"""Up-conv (4 sub-pixel phases) timing, merged launch vs the per-phase loop."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def per_phase(x, w, y, s):
    B, Hi, Wi, Ci = x.shape; k = 3; Ho = 2 * (Hi - 1) + k
    for py in range(2):
        for px in range(2):
            taps = [((py - ky) // 2, (px - kx) // 2, ky * k + kx) for ky in range(k) if (py - ky) % 2 == 0 for kx in range(k) if (px - kx) % 2 == 0]
            C.launch(x, w, y, taps, (Ho - py + 1) // 2, (Ho - px + 1) // 2, osy=2, oy0=py, ox0=px, w_tap_stride=Ci, w_row_stride=k * k * Ci, a_scale=s, precision=1)
B = 32
for ci, co, h in [(512, 512, 32), (512, 256, 64), (256, 128, 128), (512, 512, 16)]:
    x = torch.randn(B, h, h, ci, device=dev); w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    s = torch.randn(B, ci, device=dev); y = torch.empty(B, 2 * h + 1, 2 * h + 1, co, device=dev)
    fl = 2.0 * B * h * h * co * ci * 9
    m1 = timeit(lambda: C.conv_transpose2d_s2(x, w, 3, out=y, a_scale=s, precision=1))
    y1 = y.clone()
    m2 = timeit(lambda: per_phase(x, w, y, s))
    print(ci, co, h, 'merged %.1f us %.1f TF | per-phase %.1f us %.1f TF | maxdiff %.2e' % (m1 * 1e3, fl / m1 / 1e9, m2 * 1e3, fl / m2 / 1e9, (y1 - y).abs().max().item()))
