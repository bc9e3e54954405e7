"""A few fp16 / split-bf16 conv launches for rocprofv3 --pmc passes (few dispatches, bounded): the three dominant
stride-1 3x3 shapes of the StyleGAN2-256 step (patch kernel) and the 256->128 up-sampling layer (fused kernel).  usage: pmc_conv16.py [mode]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0')
B = 32
m = C.precision_code(sys.argv[1] if len(sys.argv) > 1 else 'f16')
for ci, co, h in [(512, 512, 64), (256, 256, 128), (128, 128, 256)]:
    x = torch.randn(B, h, h, ci, device=dev); w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    s = torch.randn(B, ci, device=dev); y = torch.empty(B, h, h, co, device=dev)
    ws = C.split_weight(w, m)
    for _ in range(2):
        C.conv2d(x, w, 3, pad=1, out=y, a_scale=s, precision=m, w_split=ws)
    torch.cuda.synchronize()
    print(ci, co, h, 'x MB', x.numel() * 4 / 1e6, 'y MB', y.numel() * 4 / 1e6, 'w MB', w.numel() * 4 / 1e6)
# the up-sampling layer 256->128 @128->256: fused kernel (conv_upfused.hip) in fp16 and fp16 x2
ci, co, h = 256, 128, 128
x = torch.randn(B, h, h, ci, device=dev); w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
s = torch.randn(B, ci, device=dev); dm = torch.rand(B, co, device=dev)
k1 = torch.tensor([1., 3., 3., 1.]); kern = (k1[:, None] * k1[None, :] / 64 * 4).to(dev)
nz, nw, bias = torch.randn(4 * h * h, device=dev), torch.ones(1, device=dev), torch.zeros(co, device=dev)
for mm in (2, 3):
    ws = C.split_weight(w, mm)
    for _ in range(2):
        C.upconv_blur_act(x, ws, kern, s, ci, dm, nz, nw, bias, mm)
    torch.cuda.synchronize()
