"""A few split-bf16 conv launches for rocprofv3 --pmc passes (few dispatches, bounded).  With pre-split weights the
stride-1 3x3 shapes take the patch kernel (the dominant kernel of the training step)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0')
B = 32
for ci, co, h in [(512, 512, 64), (128, 128, 256)]:
    x = torch.randn(B, h, h, ci, device=dev); w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    s = torch.randn(B, ci, device=dev); y = torch.empty(B, h, h, co, device=dev)
    ws = C.split_weight(w) if os.environ.get('PMC_NO_SPLIT') is None else None
    for _ in range(2):
        C.conv2d(x, w, 3, pad=1, out=y, a_scale=s, precision=1, w_split=ws)
    torch.cuda.synchronize()
    print(ci, co, h, 'x MB', x.numel() * 4 / 1e6, 'y MB', y.numel() * 4 / 1e6, 'w MB', w.numel() * 4 / 1e6)
