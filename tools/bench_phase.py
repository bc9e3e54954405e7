"""Phase timing of the 8-wave split-bf16 conv kernel (needs a -DWGS_ABL=20/21 build, see tools/build_abl.sh)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C
dev = torch.device('cuda:0')
B, ci, co, h = 32, 512, 512, 64
x = torch.randn(B, h, h, ci, device=dev); w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
s = torch.randn(B, ci, device=dev); y = torch.empty(B, h, h, co, device=dev)
for _ in range(3): C.conv2d(x, w, 3, pad=1, out=y, a_scale=s, precision=1)
torch.cuda.synchronize()
t = y.flatten()[:32].cpu().view(8, 4)
print(os.environ.get('WGS_LIB', 'default').split('/')[-1], 'cycles/iteration per wave [issue loads, mma(+stores), stores/none, barrier]')
for wv in range(8): print(wv, ['%.0f' % v for v in t[wv].tolist()], 'sum %.0f' % t[wv].sum().item())
