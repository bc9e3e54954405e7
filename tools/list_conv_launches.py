"""Every conv launch shape of one training step with its kernel symbol, time and rate (bench.conv_profile), heaviest first.
usage: python tools/list_conv_launches.py [precision=fp32w]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32w'
dev = torch.device('cuda:0')
bench.ENGINE_KW['two_streams'] = False
eng = bench.build(dev, 'stylegan2', 128, 32, 32, precision=prec)
for _ in range(3):
    eng.step()
recs = bench.conv_profile(eng, 2)
tot = sum(r[3] for r in recs)
for label, sym, fl, ms, n in sorted(recs, key=lambda r: -r[3]):
    print('%6.3f ms %5.1f%%  x%-4.1f %7.1f TF  %-60s %s' % (ms, 100 * ms / tot, n, fl / ms / 1e9, label, sym))
print('total conv %.2f ms' % tot)
