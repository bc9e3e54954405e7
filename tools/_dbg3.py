import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import golden_inputs as GI
from warpedganspace_amd import conv as C
from warpedganspace_amd.gan_load import StyleGAN2Wrapper
from warpedganspace_amd.stylegan2 import Generator
dev = torch.device('cuda:0')
torch.manual_seed(1)
G0 = Generator(256, 512, 8)
sd = GI.fill_state_dict(G0.state_dict(), 7000)
for k in sd:
    if k.startswith('style.') and k.endswith('weight'): sd[k] = sd[k] * 100.0
G0.load_state_dict(sd)
G = StyleGAN2Wrapper(G0.to(dev), False)
z = torch.randn(64, 512, device=dev)
with torch.no_grad():
    C.set_precision('fp32'); ref = G(z)
    C.set_precision('mixed'); img = G(z)
e = ((img - ref).abs().flatten(1).max(1).values / ref.abs().flatten(1).max(1).values).cpu()
print('MIN_RES', os.environ.get('WGS_MIXED_MIN_RES', '64'), 'UP', os.environ.get('WGS_MIXED_UP', 'f16x2'),
      'median %.2e p90 %.2e max %.2e' % (float(e.median()), float(e.kthvalue(58).values), float(e.max())), flush=True)
