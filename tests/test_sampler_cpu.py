"""CPU: the inverse-CDF truncated-normal sampler (warpedganspace_amd/aux.py::sample_z) against scipy's truncnorm, which the
reference draws from (lib/aux.py:50-53: truncnorm.rvs(-t, t, size=...)): same support, moments and distribution (KS test)."""
import numpy as np
import pytest
import torch
from scipy import stats

from warpedganspace_amd.aux import sample_z


@pytest.mark.parametrize('t', [0.5, 0.7, 1.0 + 1e-9, 2.0])
def test_truncated_sampler_matches_scipy_truncnorm(t):
    g = torch.Generator().manual_seed(123)
    z = sample_z(4000, 128, truncation=t, generator=g)
    assert z.shape == (4000, 128) and z.dtype == torch.float32
    x = z.double().numpy().ravel()
    assert x.min() >= -t and x.max() <= t
    ref = stats.truncnorm(-t, t)
    n = x.size
    assert abs(x.mean() - ref.mean()) < 5 * ref.std() / np.sqrt(n)
    assert abs(x.var() - ref.var()) < 0.01 * ref.var()
    d, pval = stats.kstest(x[:100000], ref.cdf)
    assert pval > 1e-3, (d, pval)
    # and against samples scipy itself draws (two-sample test, like the reference would produce)
    y = ref.rvs(size=100000, random_state=np.random.default_rng(5))
    assert stats.ks_2samp(x[:100000], y).pvalue > 1e-3


def test_untruncated_is_standard_normal():
    g = torch.Generator().manual_seed(1)
    z = sample_z(2000, 64, truncation=None, generator=g)
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1) < 0.02
    assert torch.equal(sample_z(3, 8, truncation=1.0, generator=torch.Generator().manual_seed(7)),
                       torch.randn(3, 8, generator=torch.Generator().manual_seed(7)))       # truncation == 1.0 means "none" (lib/aux.py:46-49)
