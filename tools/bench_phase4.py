"""Up-conv (4 sub-pixel phases) timing: merged launch (register-staged kernel) vs per-phase launches (patch form with
pre-split weights when WGS_PHASE_PATCH is set)."""
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
B = 32
for ci, co, h in [(512, 512, 32), (512, 256, 64), (256, 128, 128)]:
    x = torch.randn(B, h, h, ci, device=dev); w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    s = torch.randn(B, ci, device=dev); y = torch.empty(B, 2 * h + 1, 2 * h + 1, co, device=dev)
    ws = C.split_weight(w)
    fl = 2.0 * B * h * h * co * ci * 9
    y.zero_()
    m0 = timeit(lambda: C.conv_transpose2d_s2(x, w, 3, out=y, a_scale=s, precision=1))
    y0 = y.clone(); y.zero_()
    m1 = timeit(lambda: C.conv_transpose2d_s2(x, w, 3, out=y, a_scale=s, precision=1, w_split=ws))
    print(ci, co, h, 'no split weights %.1f us %.1f TF | with split weights %.1f us %.1f TF | maxdiff %.2e' % (m0 * 1e3, fl / m0 / 1e9, m1 * 1e3, fl / m1 / 1e9, (y0 - y).abs().max().item()))
