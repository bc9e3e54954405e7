"""GPU: wgs_sample_step — one launch draws (z, path indices, shift magnitudes) of a training step in HBM, with the distributions of the
reference's host-side sampling (lib/trainer.py:195-221, lib/aux.py:39-53) incl. its arange-weighted draw without replacement."""
import ctypes
import math

import pytest
import torch

from warpedganspace_amd import _lib as L

pytestmark = pytest.mark.gpu


def draw(dev, B, d, K, lo=0.25, hi=0.45, trunc=0.0, seed=1234, step=0):
    z = torch.empty(B, d, device=dev)
    idx = torch.empty(B, dtype=torch.int64, device=dev)
    mag = torch.empty(B, device=dev)
    L.check(L.lib().wgs_sample_step(L.ptr(z), L.ptr(idx, torch.int64), L.ptr(mag), B, d, K, L.c_float(lo), L.c_float(hi), L.c_float(trunc),
                                    ctypes.c_uint64(seed), ctypes.c_uint64(step), L.stream()), 'wgs_sample_step')
    return z, idx, mag


def test_latent_codes_are_standard_normal(dev):
    z, _, _ = draw(dev, 1024, 512, 128)
    v = z.double().flatten()
    n = v.numel()
    assert abs(float(v.mean())) < 4 / math.sqrt(n) and abs(float(v.var()) - 1) < 4 * math.sqrt(2 / n)
    assert abs(float((v ** 3).mean())) < 0.02 and abs(float((v ** 4).mean()) - 3) < 0.05
    # Kolmogorov-Smirnov distance to the normal CDF (0.1 % critical value ~ 1.95 / sqrt(n))
    s, _ = v.sort()
    cdf = 0.5 * (1 + torch.erf(s / math.sqrt(2)))
    emp = torch.arange(1, n + 1, device=dev, dtype=torch.float64) / n
    assert float((cdf - emp).abs().max()) < 1.95 / math.sqrt(n)
    # rows and columns are uncorrelated; an odd size is filled to its last element
    assert abs(float((z[:, 0].double() * z[:, 1].double()).mean())) < 0.15 and abs(float((z[0].double() * z[1].double()).mean())) < 0.2
    z2, _, _ = draw(dev, 3, 7, 5)
    assert bool(torch.isfinite(z2).all()) and float(z2.abs().min()) > 0


def test_truncated_codes(dev):
    t = 0.7
    z, _, _ = draw(dev, 512, 512, 16, trunc=t)
    assert float(z.abs().max()) <= t + 1e-6
    phi = lambda x: math.exp(-x * x / 2) / math.sqrt(2 * math.pi)
    Z = math.erf(t / math.sqrt(2))
    var = 1 - 2 * t * phi(t) / Z                     # variance of the standard normal truncated to [-t, t]
    assert abs(float(z.double().var()) - var) < 0.01 and abs(float(z.double().mean())) < 0.01
    # truncation 1.0 means "no truncation" in the reference (aux.py:49)
    z1, _, _ = draw(dev, 512, 512, 16, trunc=1.0)
    assert float(z1.abs().max()) > 3.0


def test_path_indices_are_uniform(dev):
    K = 128
    counts = torch.zeros(K, dtype=torch.float64, device=dev)
    for s in range(40):
        _, idx, _ = draw(dev, 1024, 8, K, step=s)
        assert int(idx.min()) >= 0 and int(idx.max()) < K
        counts += torch.bincount(idx, minlength=K).double()
    e = counts.sum() / K
    chi2 = float(((counts - e) ** 2 / e).sum())
    assert chi2 < K - 1 + 5 * math.sqrt(2 * (K - 1)), chi2          # mean K-1, five standard deviations
    _, idx3, _ = draw(dev, 1000, 8, 3)
    assert set(idx3.tolist()) == {0, 1, 2}


def test_shift_magnitudes_follow_the_reference_draw(dev):
    """B of the 2B pool entries without replacement with weights 0 .. 2B-1: ranges of the two halves, no entry twice, and the number
    of positive magnitudes per batch distributed as torch.multinomial's draw from the same weights (mean and variance)."""
    B, lo, hi, T = 32, 0.25, 0.45, 600
    npos = []
    for s in range(T):
        _, _, mag = draw(dev, B, 8, 16, lo, hi, step=s, seed=77)
        m = mag.tolist()
        assert all((lo <= v <= hi) or (-hi <= v <= -lo) for v in m), m
        assert len(set(m)) >= B - 1          # (24-bit uniforms: two pool entries coincide once in ~10^4 batches)
        npos.append(sum(v > 0 for v in m))
    g = torch.Generator(device=dev).manual_seed(5)
    w = torch.arange(2 * B, dtype=torch.float, device=dev)
    ref = [(torch.multinomial(w, B, replacement=False, generator=g) >= B).sum().item() for _ in range(T)]
    mean, rmean = sum(npos) / T, sum(ref) / T
    var = sum((v - mean) ** 2 for v in npos) / T
    rvar = sum((v - rmean) ** 2 for v in ref) / T
    print('positive magnitudes per batch of %d: kernel %.2f +- %.2f, torch.multinomial %.2f +- %.2f' % (B, mean, var ** 0.5, rmean, rvar ** 0.5))
    assert 0.64 < mean / B < 0.78 and abs(mean - rmean) < 5 * (rvar / T) ** 0.5 * 2 ** 0.5 + 0.05
    assert 0.6 < var / rvar < 1.6
    # entry 0 of the pool (weight 0) is never drawn even when every other entry is: B = 1 pool {neg_0, pos_0} -> always pos_0
    for s in range(50):
        _, _, m1 = draw(dev, 1, 4, 2, lo, hi, step=s)
        assert float(m1[0]) >= lo


def test_stateless_and_reproducible(dev):
    a = draw(dev, 32, 512, 128, seed=9, step=3)
    b = draw(dev, 32, 512, 128, seed=9, step=3)
    c = draw(dev, 32, 512, 128, seed=9, step=4)
    d = draw(dev, 32, 512, 128, seed=10, step=3)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert not torch.equal(a[0], c[0]) and not torch.equal(a[0], d[0]) and not torch.equal(a[2], c[2])
    assert abs(float((a[0].double() * c[0].double()).mean())) < 0.05 and abs(float((a[0].double() * d[0].double()).mean())) < 0.05
    with pytest.raises(L.WgsError):
        draw(dev, 2048, 8, 4)
