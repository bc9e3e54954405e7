"""CPU: the oracle restatements reproduce the golden vectors generated from the reference modules
(tools/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import wgs_oracle as O
from tests import golden_inputs as GI
from tests.util import rel_err

SS_CASES = {'tiny': (4, 2, 8, 3, 11), 'cfg1': (32, 8, 128, 4, 12), 'cfg3': (128, 32, 512, 4, 13),
            'cfg4': (16, 4, 120, 5, 14)}


@pytest.mark.parametrize('name', list(SS_CASES))
@pytest.mark.parametrize('lg', [True, False])
def test_support_sets_torch_oracle_vs_reference(golden, name, lg):
    g = golden('support_sets')
    K, N, d, B, seed = SS_CASES[name]
    c = GI.support_sets_case(K, N, d, B, seed, learn_gammas=lg)
    sd = {k: v.clone().requires_grad_(True) for k, v in c['sd'].items()}
    z = c['z'].clone().requires_grad_(True)
    y = O.support_sets_forward(sd, GI.one_hot(c['idx'], K), z, lg, c['gamma'])
    (y * c['gout']).sum().backward()
    tag = '%s_%s' % (name, 'lg' if lg else 'cg')
    assert rel_err(y, g[tag + '_out']) < 1e-6
    assert rel_err(z.grad, g[tag + '_dz']) < 1e-5
    assert rel_err(sd['ALPHAS'].grad, g[tag + '_dalphas']) < 1e-5
    if lg:
        assert rel_err(sd['LOGGAMMA'].grad, g[tag + '_dloggamma']) < 1e-4


@pytest.mark.parametrize('name', list(SS_CASES))
@pytest.mark.parametrize('lg', [True, False])
def test_support_sets_c_oracle_vs_reference(golden, name, lg):
    """The plain-C fp64 restatement (forward + analytic backward) against the reference's autograd."""
    g = golden('support_sets')
    K, N, d, B, seed = SS_CASES[name]
    c = GI.support_sets_case(K, N, d, B, seed, learn_gammas=lg)
    tag = '%s_%s' % (name, 'lg' if lg else 'cg')
    out, _ = O.rbf_c_forward(c['sd'], c['idx'], c['z'], lg, c['gamma'])
    assert rel_err(out, g[tag + '_out']) < 2e-6
    dtable, dal, dlg, dz = O.rbf_c_backward(c['sd'], c['idx'], c['z'], c['gout'], lg, c['gamma'])
    assert rel_err(dz, g[tag + '_dz']) < 2e-5
    assert rel_err(dal, g[tag + '_dalphas']) < 2e-5
    if lg:
        # d/dloggamma sums terms ~ (1 - gamma*r2) with gamma*r2 ~ 1: the reference's own fp32 autograd
        # carries ~1e-4 cancellation noise relative to this fp64 evaluation.
        assert rel_err(dlg.reshape(-1, 1), g[tag + '_dloggamma']) < 5e-4
    if tag + '_dtable' in g:
        assert rel_err(dtable, g[tag + '_dtable']) < 2e-5
    else:
        rows = torch.unique(c['idx']).numpy()
        assert rel_err(dtable[rows][:, ::16], g[tag + '_dtable_rows_sub16']) < 2e-5
        vn = np.linalg.norm(dtable[rows].reshape(len(rows), 2 * N, d), axis=2)
        assert rel_err(vn, g[tag + '_dtable_vecnorm']) < 2e-5
        rest = np.delete(dtable, rows, axis=0)
        assert np.abs(rest).max() == 0.0 and g[tag + '_dtable_rest_absmax'] == 0.0


def test_traverse_oracle_vs_reference(golden):
    g = golden('support_sets')
    c = GI.support_sets_case(6, 3, 16, 2, 21, learn_gammas=True)
    path, _ = O.traverse_paths(c['sd'], c['z'], 0.2, 3, True, c['gamma'])
    assert rel_err(path, g['traverse_path']) < 1e-6


def test_upfirdn2d_oracle_vs_reference_native(golden):
    g = golden('native_ops')
    k = O.make_blur_kernel()
    for i, c in enumerate(GI.UPFIRDN_CASES):
        x = GI.rt(100 + i, c['major'], c['h'], c['w'], c['minor'])
        kk = k * c['gain'] if c['name'] != 'up3_down2' else GI.rt(777, 5, 3)
        y = O.upfirdn2d_mhwc(x, kk, c['up'], c['up'], c['down'], c['down'], *c['pad'])
        ref = g['upfirdn_' + c['name']]
        assert tuple(y.shape) == ref.shape, c['name']
        assert rel_err(y, ref) < 1e-6, c['name']
