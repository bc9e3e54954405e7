"""Weight-gradient kernel (exact fp32) on the ResNet-18 layer shapes of the bench (B=32 pairs)."""
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
tot = 0.0
for ci, co, h, s in [(64, 64, 64, 1), (64, 128, 64, 2), (128, 128, 32, 1), (128, 256, 32, 2), (256, 256, 16, 1), (256, 512, 16, 2), (512, 512, 8, 1)]:
    ho = (h + 2 - 3) // s + 1
    x = torch.randn(B, h, h, ci, device=dev); g = torch.randn(B, ho, ho, co, device=dev)
    dw = torch.zeros(co, 9, ci, device=dev)
    fl = 2.0 * B * ho * ho * co * ci * 9
    t0 = timeit(lambda: C.conv2d_wgrad(x, g, dw, 3, stride=s, pad=1))
    tot += t0
    print(os.environ.get('WGS_LIB', 'default').split('/')[-1], ci, co, h, s, 'fp32 wgrad %.1f us %.1f TF' % (t0 * 1e3, fl / t0 / 1e9))
print('sum %.1f us' % (tot * 1e3))
