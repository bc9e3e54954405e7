"""cProfile of the host side of the training step (the queue running, no synchronisation inside the profiled region): where the ~7 ms of
enqueue time per step go.  usage: python tools/host_cprofile.py [precision] [steps]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'auto'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
eng = bench.build(dev, 'stylegan2', 128, 32, 32, precision=prec)
for _ in range(10):
    eng.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
reps = max(1, steps // 2)
for _ in range(reps):            # two steps at a time into an EMPTY queue: the launches never wait for a queue slot
    torch.cuda.synchronize()
    pr.enable()
    eng.step(); eng.step()
    pr.disable()
torch.cuda.synchronize()
steps = 2 * reps
for key in ('tottime', 'cumulative'):
    print('=' * 20, key, '(totals over %d steps; cProfile roughly doubles the host time)' % steps)
    pstats.Stats(pr).strip_dirs().sort_stats(key).print_stats(35)
