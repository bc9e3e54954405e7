"""Two launches of the LDS-DMA conv kernel fed with a producer-written fp16 plane (wgs_conv_desc.x_f16) for rocprofv3 --pmc passes:
the stride-1 gradient convs 512->512 @64x64 and 256->256 @128x128 (B = 32) as the generator backward issues them.  usage: pmc_dma16.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0')
B = 32
for ch, h in [(512, 64), (256, 128)]:
    plane = (torch.randn(B, h, h, ch, device=dev) * 1000.0).half().view(torch.int16)
    wt = torch.randn(9, ch, ch, device=dev) / (9 * ch) ** 0.5
    wts = C.split_weight(wt, 2)
    am = torch.full((1,), 3.0, device=dev)
    for _ in range(3):
        g = C.conv2d_dgrad(plane, wt, (h, h), 3, pad=1, w_split=wts, a_amax=am, a_bound=1.0, precision=2, x_f16=True)
    torch.cuda.synchronize()
    print(ch, h, float(g.abs().max()))
