"""Where the host thread spends a training step: perf_counter marks at the step's phases (monkey-patched around the engine's calls), with
the queue running (no synchronisation between steps).  If the host is far ahead of the device, the marks of one step span the enqueue
time (~6 ms) and consecutive steps start ~6 ms apart; a host that waits for the device somewhere shows the wait at that mark.
usage: python tools/host_marks.py [precision]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from warpedganspace_amd import trainer as T  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'auto'
dev = torch.device('cuda:0')
eng = bench.build(dev, 'stylegan2', 128, 32, 32, precision=prec)
for _ in range(10):
    eng.step()
torch.cuda.synchronize()
marks = []


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        marks.append((label, t0, time.perf_counter()))
        return r
    setattr(obj, name, g)


wrap(eng, 'sample', 'sample')
wrap(eng.G, 'begin', 'G.begin')
wrap(eng.G, 'advance', 'G.advance')
wrap(eng.G, 'finish', 'G.finish')
wrap(eng.R, '_forward_impl', 'R.fwd')
wrap(eng.R, '_backward_impl', 'R.bwd')
wrap(eng.bucket, 'adam_step', 'adam')
wrap(eng.bucket, 'zero_grad', 'zero_grad')
orig_G_call = eng.G.forward
def gfwd(*a, **k):
    t0 = time.perf_counter(); r = orig_G_call(*a, **k); marks.append(('G.forward', t0, time.perf_counter())); return r
eng.G.forward = gfwd
steps = []
t_all = time.perf_counter()
for i in range(12):
    t0 = time.perf_counter()
    eng.step()
    steps.append((t0, time.perf_counter()))
t_host = time.perf_counter() - t_all
torch.cuda.synchronize()
t_dev = time.perf_counter() - t_all
print('12 steps: host loop %.1f ms, device done after %.1f ms' % (1e3 * t_host, 1e3 * t_dev))
base = steps[0][0]
for i, (a, b) in enumerate(steps):
    print('step %2d: host start +%.2f ms, host time %.2f ms' % (i, 1e3 * (a - base), 1e3 * (b - a)))
    for label, m0, m1 in marks:
        if a <= m0 <= b and i in (5, 6):
            print('      %-10s +%.2f .. +%.2f ms (%.2f)' % (label, 1e3 * (m0 - a), 1e3 * (m1 - a), 1e3 * (m1 - m0)))
