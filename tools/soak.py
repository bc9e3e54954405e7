#!/usr/bin/env python
"""Soak run of the training step at the headline size (StyleGAN2-256, K=128, N=32, B=32) in the default arithmetic: N iterations,
statistics popped every 250, everything must stay finite and the run-time precision check must keep passing as R and S evolve.
usage: python tools/soak.py [iterations=3000] [precision=auto]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
prec = sys.argv[2] if len(sys.argv) > 2 else 'auto'
dev = torch.device('cuda:0')
eng = bench.build(dev, 'stylegan2', 128, 32, 32, precision=prec)
t0 = time.time()
for it in range(1, n + 1):
    eng.step()
    if it % 250 == 0:
        st = eng.pop_stats()
        chk = eng.check_precision()
        finite = bool(torch.isfinite(eng.bucket.flat).all()) and bool(torch.isfinite(eng.bucket.exp_avg_sq).all())
        print('iter %5d  %.1f it/s  acc %.3f ce %.4f l1 %.4f  params finite %s  precision check %s' % (
            it, it / (time.time() - t0), st['accuracy'], st['classification_loss'], st['regression_loss'], finite,
            'n/a' if chk is None else '%.2e (%s)' % (chk['batch'], 'ok' if chk['ok'] else 'OVER')), flush=True)
        assert finite and all(v == v for v in st.values()) and (chk is None or chk['ok'])
print('soak ok: %d iterations in %.1f s' % (n, time.time() - t0))
