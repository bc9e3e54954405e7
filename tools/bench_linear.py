"""Micro-benchmark of the small dense layers of the mapping / modulation path (M = 32 rows)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import _lib as L
dev = torch.device('cuda:0')
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
lib, st = L.lib(), L.stream()
for M, N, K in [(32, 512, 512), (32, 5000, 512), (64, 512, 512)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.zeros(N, device=dev); y = torch.empty(M, N, device=dev)
    g = torch.randn(M, N, device=dev); gx = torch.empty(M, K, device=dev)
    t1 = timeit(lambda: L.check(lib.wgs_linear_fwd(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), M, N, K, K, N, L.c_float(1.0), L.c_float(1.0), 0, 1, L.c_float(0.0), L.c_float(1.0), st), 'f'))
    t2 = timeit(lambda: L.check(lib.wgs_linear_dgrad(L.ptr(g), L.ptr(w), L.ptr(y), L.ptr(gx), M, N, K, N, K, L.c_float(1.0), L.c_float(0.2), L.c_float(1.41), 0, st), 'd'))
    print('M=%d N=%d K=%d: fwd %.1f us, dgrad %.1f us' % (M, N, K, t1, t2))
# the mapping network: 1 + 8 launches vs one launch
import ctypes
for B in (32, 64):
    d, nl = 512, 8
    z = torch.randn(B, d, device=dev); ws = [torch.randn(d, d, device=dev) for _ in range(nl)]; bs = [torch.zeros(d, device=dev) for _ in range(nl)]
    acts = torch.empty(nl + 1, B, d, device=dev)
    wp = (ctypes.c_void_p * nl)(*[w.data_ptr() for w in ws]); bp = (ctypes.c_void_p * nl)(*[b.data_ptr() for b in bs])
    def per_layer():
        L.check(lib.wgs_pixelnorm_fwd(L.ptr(z), L.ptr(acts[0]), B, d, L.c_float(1e-8), st), 'pn')
        for l in range(nl):
            L.check(lib.wgs_linear_fwd(L.ptr(acts[l]), L.ptr(ws[l]), L.ptr(bs[l]), L.ptr(acts[l + 1]), B, d, d, d, d, L.c_float(0.01), L.c_float(0.01), 0, 1, L.c_float(0.0), L.c_float(1.0), st), 'l')
    t1 = timeit(per_layer)
    t2 = timeit(lambda: L.check(lib.wgs_mapping_mlp_fwd(L.ptr(z), wp, bp, L.ptr(acts), B, d, nl, L.c_float(0.01), L.c_float(0.01), L.c_float(1e-8), st), 'm'))
    print('mapping network B=%d: 9 launches %.1f us, one launch %.1f us' % (B, t1, t2))
