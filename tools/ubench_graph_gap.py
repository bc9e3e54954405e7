#!/usr/bin/env python
"""Micro-benchmark: GPU time per kernel of a chain of N dependent launches on one stream, enqueued one by one vs replayed from a captured
HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm).  The kernels are long enough (~20 us) for the host to stay ahead in the eager form, so the
difference is the device-side cost of a kernel boundary.  usage: python tools/ubench_graph_gap.py"""
import torch

dev = torch.device('cuda:0')
N = 400


def run(n_elem):
    x = torch.zeros(n_elem, device=dev)

    def chain():
        for _ in range(N):
            x.add_(1.0)

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e))
        return best * 1e3 / N

    t_eager = timed(chain)
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        chain(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            chain()
    t_graph = timed(g.replay)
    print('%9d floats: eager %.2f us / kernel, graph %.2f us / kernel' % (n_elem, t_eager, t_graph), flush=True)


for n in (1 << 10, 1 << 20, 1 << 22, 1 << 23, 1 << 24):
    run(n)
