#!/usr/bin/env python
"""One stride-1 3x3 conv shape through the LDS-DMA form (pre-pass + igemm_dma16_kernel), fp16 operands; run under
rocprofv3 --kernel-trace --stats to get the conv kernel's own time.  usage: WGS_DMA_ALWAYS=1 WGS_NO_PATCH=1 python tools/bench_dma16.py ci co h [mode]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C

dev = torch.device('cuda:0')
ci, co, h = (int(v) for v in sys.argv[1:4])
m = C.precision_code(sys.argv[4] if len(sys.argv) > 4 else 'f16')
B = 32
torch.manual_seed(0)
x = torch.randn(B, h, h, ci, device=dev)
w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
s = torch.randn(B, ci, device=dev); dm = torch.rand(B, co, device=dev)
y = torch.empty(B, h, h, co, device=dev)
nz, nw, bias = torch.randn(h * h, device=dev), torch.ones(1, device=dev), torch.zeros(co, device=dev)
ws = C.split_weight(w, m)
kw = dict(a_scale=s, col_scale=dm, noise=nz, noise_w=nw, bias=bias, act_slope=0.2, gain=1.41, precision=m, w_split=ws)
ref = None
for i in range(12):
    C.conv2d(x, w, 3, pad=1, out=y, **kw)
    if i == 0:
        ref = y.clone()
    elif not torch.equal(ref, y):
        print('MISMATCH at launch', i, (ref - y).abs().max().item()); break
torch.cuda.synchronize()
s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s_.record()
for _ in range(8):
    C.conv2d(x, w, 3, pad=1, out=y, **kw)
e_.record(); torch.cuda.synchronize()
ms = s_.elapsed_time(e_) / 8
import hashlib
print('dma form %d->%d @%d %s: pre-pass + conv %.3f ms (%.1f TF incl. pre-pass)  sha %s' % (ci, co, h, C.precision_name(m), ms, 2.0 * B * h * h * co * ci * 9 / ms / 1e9, hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]), flush=True)
