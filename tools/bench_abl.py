import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0')
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B=32
m = C.precision_code(sys.argv[1] if len(sys.argv) > 1 else 'f16')
for ci, co, h in [(512,512,64),(256,256,128),(128,128,256)]:
    x = torch.randn(B, h, h, ci, device=dev); w = torch.randn(co, 9, ci, device=dev) / (9*ci)**0.5
    s = torch.randn(B, ci, device=dev); y = torch.empty(B, h, h, co, device=dev)
    ws = C.split_weight(w, m)
    ms = timeit(lambda: C.conv2d(x, w, 3, pad=1, out=y, a_scale=s, precision=m, w_split=ws))
    fl = 2.0*B*h*h*co*ci*9
    print(os.environ.get('WGS_LIB','default').split('/')[-1], ci, co, h, '%.3f ms %.1f TF' % (ms, fl/ms/1e9), flush=True)
