import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpedganspace_amd import conv as C
from warpedganspace_amd import _lib as L
dev = torch.device('cuda:0')
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
torch.manual_seed(0)
for B, ci, co, h in [(2, 512, 512, 64), (2, 256, 256, 128), (2, 128, 128, 256), (2, 512, 512, 32), (2, 512, 512, 16), (32, 128, 128, 256)]:
    w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
    wt = C.repack_w_t(w, co, 9, ci)
    for mag in (1.0, 1e-6):
        g = torch.randn(B, h, h, co, device=dev) * mag
        am = g.abs().max().reshape(1)
        ref = C.conv2d_dgrad(g, wt, (h, h), 3, pad=1, precision=1)
        ws = C.split_weight(wt, 2)
        out = []
        for flag in ('', '1'):
            if flag: os.environ['WGS_PATCH_TPS1'] = '1'
            else: os.environ.pop('WGS_PATCH_TPS1', None)
            L.lib().wgs_dev_reload_flags()
            for rep in range(3):
                d = C.conv2d_dgrad(g, wt, (h, h), 3, pad=1, precision=2, w_split=ws, a_amax=am)
                out.append('%s%.1e%s' % ('T1:' if flag else 'T3:', rel(d, ref), '!' if torch.isnan(d).any() else ''))
        print(B, ci, co, h, 'mag', mag, ' '.join(out), flush=True)
