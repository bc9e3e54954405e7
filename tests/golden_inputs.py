"""Closed-form / seeded input recipes shared by tools/make_golden.py (runs in the build container,
imports the reference) and the tests (run anywhere, never touch the reference).  Only OUTPUTS are
stored in tests/golden/*.npz; inputs and weights are regenerated from numpy.random.default_rng(seed)
on both sides."""
import math

import numpy as np
import torch


def rt(seed, *shape, scale=1.0):
    """float32 torch tensor of N(0, scale^2) from numpy's PCG64 (machine independent)."""
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def support_sets_case(K, N, d, B, seed, learn_gammas=True):
    """Parameters follow the reference's init law (lib/support_sets.py:39-70) but are drawn from
    numpy so both sides agree; then perturbed so dipoles are no longer exactly antipodal (all 2N
    vectors are free parameters after the first optimiser step)."""
    rng = np.random.default_rng(seed)
    radii = 1.0 + 3.0 * np.arange(K) / K
    v = rng.standard_normal((K, N, d))
    v /= np.linalg.norm(v, axis=2, keepdims=True)
    sv = np.stack([v, -v], axis=2).reshape(K, 2 * N, d) * radii[:, None, None]
    sv += 0.05 * rng.standard_normal(sv.shape)
    alphas = np.tile(np.array([1.0, -1.0]), N)[None, :].repeat(K, 0) * (1.0 + 0.1 * rng.standard_normal((K, 2 * N)))
    gamma = 1.0 / d
    loggamma = math.log(gamma) + 0.2 * rng.standard_normal((K, 1))
    idx = rng.integers(0, K, size=B)
    if B >= 2:
        idx[1] = idx[0]  # force a duplicate path index in the batch
    z = rng.standard_normal((B, d))
    gout = rng.standard_normal((B, d))
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    sd = {'SUPPORT_SETS': f32(sv.reshape(K, 2 * N * d)), 'ALPHAS': f32(alphas), 'LOGGAMMA': f32(loggamma)}
    return dict(sd=sd, idx=torch.from_numpy(idx.astype(np.int64)), z=f32(z), gout=f32(gout), gamma=gamma,
                learn_gammas=learn_gammas, K=K, N=N, d=d, B=B)


def one_hot(idx, K):
    m = torch.zeros(idx.shape[0], K)
    m[torch.arange(idx.shape[0]), idx] = 1.0
    return m


def fill_state_dict(sd, seed, skip_suffixes=('kernel', 'num_batches_tracked'), fan_in=False, per_key=False):
    """Deterministically overwrite every float tensor of a state_dict (in key order) with seeded
    normal values of a magnitude that keeps activations O(1). Returns a new dict."""
    import zlib
    rng = np.random.default_rng(seed)
    out = {}
    for k, v in sd.items():
        if any(k.endswith(s) for s in skip_suffixes) or not torch.is_floating_point(v):
            out[k] = v.clone()
            continue
        if per_key:      # independent of the order in which the module registers its tensors
            rng = np.random.default_rng([seed, zlib.crc32(k.encode())])
        a = rng.standard_normal(tuple(v.shape)).astype(np.float32)
        if k.endswith('running_var') or k.endswith('stored_var'):
            a = np.abs(a) + 0.5
        elif k.endswith('bias') or k.endswith('.b') or k.endswith('running_mean') or k.endswith('stored_mean'):
            a = a * 0.1
        elif 'modulation.bias' in k:
            a = 1.0 + a * 0.1
        elif k.endswith('noise.weight'):
            a = a * 0.1
        elif k.endswith('.scale'):
            a = np.abs(a) * 0.05 + 0.02
        elif fan_in and k.endswith('weight') and v.dim() >= 2:      # keep activations O(1) (no tanh saturation)
            a = a * (2.0 / float(np.prod(v.shape[1:]))) ** 0.5
        elif fan_in and k.endswith('weight') and v.dim() == 1:      # BatchNorm gains around 1
            a = 1.0 + 0.2 * a
        out[k] = torch.from_numpy(a).reshape(v.shape)
    return out


UPFIRDN_CASES = [
    # (major, h, w, minor, up, down, pad_x0, pad_x1, pad_y0, pad_y1, gain) — modes reached by the generator
    dict(name='blur_after_convT', major=3, h=9, w=9, minor=4, up=1, down=1, pad=(1, 1, 1, 1), gain=4.0),
    dict(name='skip_upsample', major=6, h=8, w=8, minor=1, up=2, down=1, pad=(2, 1, 2, 1), gain=4.0),
    dict(name='skip_upsample_bwd', major=6, h=16, w=16, minor=1, up=1, down=2, pad=(1, 2, 1, 2), gain=4.0),
    dict(name='ragged_crop', major=2, h=7, w=5, minor=3, up=1, down=1, pad=(-1, 2, 0, 1), gain=1.0),
    dict(name='up3_down2', major=2, h=5, w=6, minor=2, up=3, down=2, pad=(2, 2, 1, 3), gain=1.0),
]
