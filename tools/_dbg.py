import sys; sys.path.insert(0,'/root/repo')
import torch
from warpedganspace_amd import conv as C
dev=torch.device('cuda:0')
torch.manual_seed(0)
B,Ci,Co,H=4,64,256,128
x=torch.randn(B,H,H,Ci,device=dev)
wp=C.pack_weight(torch.randn(Co,Ci,3,3)/(Ci*9)**0.5).to(dev)
ws=C.split_weight(wp)
y1=C.conv2d(x,wp,3,pad=1,precision=1,w_split=ws)
y0=C.conv2d(x,wp,3,pad=1,precision=1)
d=(y1-y0).abs()
print('maxdiff',d.max().item(),'ref max',y0.abs().max().item(), 'nonzero frac', (d>0).float().mean().item())
# which positions
idx=(d>1e-3).nonzero()
print(idx[:10], idx.shape)
yx=C.conv2d(x,wp,3,pad=1,precision=0)
print('dma vs fp32', ((y1-yx).abs().max()/yx.abs().max()).item(), 'reg vs fp32', ((y0-yx).abs().max()/yx.abs().max()).item())
s=torch.randn(B,Ci,device=dev)+1.0
y1=C.conv2d(x,wp,3,pad=1,precision=1,w_split=ws,a_scale=s)
y0=C.conv2d(x,wp,3,pad=1,precision=1,a_scale=s)
d=(y1-y0).abs()
print('style: maxdiff',d.max().item(),'ref max',y0.abs().max().item(), 'nonzero frac', (d>0).float().mean().item())
idx=(d>0).nonzero(); print(idx[:5], idx.shape)
dm=torch.rand(B,Co,device=dev)+0.5
y1=C.conv2d(x,wp,3,pad=1,precision=1,w_split=ws,col_scale=dm)
y0=C.conv2d(x,wp,3,pad=1,precision=1,col_scale=dm)
print('demod only: maxdiff',(y1-y0).abs().max().item())
s2=torch.full((B,Ci),2.0,device=dev); s2[:, ::2]=0.5
y1=C.conv2d(x,wp,3,pad=1,precision=1,w_split=ws,a_scale=s2)
y0=C.conv2d(x,wp,3,pad=1,precision=1,a_scale=s2)
print('pow2 style: maxdiff',(y1-y0).abs().max().item())
s3=s.clone(); s3[1:]=s3[:1]
y1=C.conv2d(x,wp,3,pad=1,precision=1,w_split=ws,a_scale=s3)
y0=C.conv2d(x,wp,3,pad=1,precision=1,a_scale=s3)
d=(y1-y0).abs(); print('same style all samples: maxdiff',d.max().item(), [d[b].max().item() for b in range(B)])
y1=C.conv2d(x,wp,3,pad=1,precision=1,w_split=ws,a_scale=s)
y0=C.conv2d(x,wp,3,pad=1,precision=1,a_scale=s)
d=(y1-y0).abs(); print('per-sample maxdiff', [d[b].max().item() for b in range(B)])
