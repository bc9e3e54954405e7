"""Small generator layers (few tiles): split-bf16 conv with and without split-K."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B = 32
for ci, co, h in [(512, 512, 4), (512, 512, 8), (512, 512, 16)]:
    x = torch.randn(B, h, h, ci, device=dev); w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    s = torch.randn(B, ci, device=dev); y = torch.empty(B, h, h, co, device=dev)
    ms = timeit(lambda: C.conv2d(x, w, 3, pad=1, out=y, a_scale=s, precision=1))
    yt = torch.empty(B, 2 * h + 1, 2 * h + 1, co, device=dev)
    ms2 = timeit(lambda: C.conv_transpose2d_s2(x, w, 3, out=yt, a_scale=s, precision=1))
    fl = 2.0 * B * h * h * co * ci * 9
    print(os.environ.get('WGS_LIB', 'default').split('/')[-1], ci, co, h, 'conv %.1f us %.1f TF | convT(4 phases) %.1f us %.1f TF' % (ms * 1e3, fl / ms / 1e9, ms2 * 1e3, fl / ms2 / 1e9))
