"""GPU: HIP RBF kernels (through the C ABI) vs the oracle and vs the reference golden vectors."""
import numpy as np
import pytest
import torch

from oracle import wgs_oracle as O
from tests import golden_inputs as GI
from tests.util import rel_err
from warpedganspace_amd.support_sets import SupportSets

pytestmark = pytest.mark.gpu

SS_CASES = {'tiny': (4, 2, 8, 3, 11), 'cfg1': (32, 8, 128, 4, 12), 'cfg3': (128, 32, 512, 4, 13),
            'cfg4': (16, 4, 120, 5, 14)}
TOL = 1e-5  # fp32 kernel vs fp64 oracle; north_star gate is 1e-3


def build(c, dev, learn_alphas=True):
    S = SupportSets(c['K'], c['N'], c['d'], learn_alphas=learn_alphas, learn_gammas=c['learn_gammas'], gamma=c['gamma'])
    S.load_state_dict(c['sd'])
    return S.to(dev)


@pytest.mark.parametrize('name', list(SS_CASES))
@pytest.mark.parametrize('lg', [True, False])
def test_rbf_fwd_bwd_vs_golden_and_oracle(dev, golden, name, lg):
    g = golden('support_sets')
    K, N, d, B, seed = SS_CASES[name]
    c = GI.support_sets_case(K, N, d, B, seed, learn_gammas=lg)
    S = build(c, dev)
    z = c['z'].to(dev).requires_grad_(True)
    y = S(GI.one_hot(c['idx'], K).to(dev), z)
    (y * c['gout'].to(dev)).sum().backward()
    tag = '%s_%s' % (name, 'lg' if lg else 'cg')
    # vs reference golden (fp32 torch autograd)
    assert rel_err(y, g[tag + '_out']) < 1e-5
    assert rel_err(z.grad, g[tag + '_dz']) < 1e-4
    assert rel_err(S.ALPHAS.grad, g[tag + '_dalphas']) < 1e-4
    if lg:
        assert rel_err(S.LOGGAMMA.grad, g[tag + '_dloggamma']) < 1e-3
    else:
        assert S.LOGGAMMA.grad is None
    # vs fp64 C oracle
    out, _ = O.rbf_c_forward(c['sd'], c['idx'], c['z'], lg, c['gamma'])
    dtable, dal, dlg, dz = O.rbf_c_backward(c['sd'], c['idx'], c['z'], c['gout'], lg, c['gamma'])
    assert rel_err(y, out) < TOL
    assert rel_err(S.SUPPORT_SETS.grad, dtable) < TOL * 5
    assert rel_err(z.grad, dz) < TOL * 5
    assert rel_err(S.ALPHAS.grad, dal) < TOL * 5
    if lg:
        assert rel_err(S.LOGGAMMA.grad.reshape(-1), dlg) < 5e-4
    # rows that were not selected get exactly zero gradient (dense [K, 2N*d] grad, row-sparse content)
    sel = set(c['idx'].tolist())
    rest = [k for k in range(K) if k not in sel]
    if rest:
        assert S.SUPPORT_SETS.grad[rest].abs().max().item() == 0.0
    assert torch.allclose(y.norm(dim=1), torch.ones(B, device=dev), atol=1e-5)


@pytest.mark.parametrize('K,N,d,B', [(128, 32, 512, 32), (64, 16, 512, 32), (200, 64, 512, 8), (32, 8, 128, 16),
                                       (7, 3, 36, 5), (5, 1, 1024, 2), (3, 2, 2048, 1)])
def test_rbf_config_shapes_vs_oracle(dev, K, N, d, B):
    """All BASELINE config shapes (cfg5's 262 KB row included) + ragged shapes, with fused scale."""
    c = GI.support_sets_case(K, N, d, B, 1000 + K + d, learn_gammas=True)
    S = build(c, dev, learn_alphas=False)
    mag = (torch.rand(B) * 0.2 + 0.25) * torch.where(torch.rand(B) > 0.5, 1.0, -1.0)
    y = S.forward_idx(c['idx'].to(dev), c['z'].to(dev), scale=mag.to(dev))
    (y * c['gout'].to(dev)).sum().backward()
    out, _ = O.rbf_c_forward(c['sd'], c['idx'], c['z'], True, c['gamma'])
    assert rel_err(y, out * mag.double().numpy()[:, None]) < TOL
    dtable, _, dlg, _ = O.rbf_c_backward(c['sd'], c['idx'], c['z'], c['gout'] * mag[:, None], True, c['gamma'])
    assert rel_err(S.SUPPORT_SETS.grad, dtable) < TOL * 5
    assert rel_err(S.LOGGAMMA.grad.reshape(-1), dlg) < 1e-3
    assert S.ALPHAS.grad is None


def test_rbf_full_size_properties(dev):
    """cfg3 full size: unit norm; antipodal symmetry f(-z; flipped set) = -f(z); scale linearity."""
    K, N, d, B = 128, 32, 512, 32
    S = SupportSets(K, N, d, learn_gammas=True).to(dev)
    z = torch.randn(B, d, device=dev)
    idx = torch.randint(0, K, (B,), device=dev)
    y = S.forward_idx(idx, z)
    assert torch.allclose(y.norm(dim=1), torch.ones(B, device=dev), atol=1e-5)
    # at init the support set is symmetric under s -> -s with alpha -> -alpha, so f(-z) = f(z)... check
    # the generic identity instead: field of the negated set at -z is minus the field at z.
    S2 = SupportSets(K, N, d, learn_gammas=True).to(dev)
    S2.load_state_dict(S.state_dict())
    with torch.no_grad():
        S2.SUPPORT_SETS.neg_()
    y2 = S2.forward_idx(idx, -z)
    assert rel_err(y2, -y) < 1e-5
    y3 = S.forward_idx(idx, z, scale=torch.full((B,), 0.3, device=dev))
    assert rel_err(y3, 0.3 * y) < 1e-6


def test_traverse_vs_golden_and_oracle(dev, golden):
    g = golden('support_sets')
    c = GI.support_sets_case(6, 3, 16, 2, 21, learn_gammas=True)
    S = build(c, dev, learn_alphas=False)
    path, shift = S.traverse(c['z'].to(dev), 0.2, 3)
    assert rel_err(path, g['traverse_path']) < 1e-5
    opath, oshift = O.traverse_paths(c['sd'], c['z'], 0.2, 3, True, c['gamma'])
    assert rel_err(shift, oshift) < 1e-5
    # LDS-resident (cfg3: 131 KB set) and global-fallback (cfg5: 262 KB set) variants agree with
    # step-by-step forward calls
    for (K, N, d) in ((8, 32, 512), (4, 64, 512)):
        c = GI.support_sets_case(K, N, d, 2, 77, learn_gammas=True)
        S = build(c, dev, learn_alphas=False)
        path, shift = S.traverse(c['z'].to(dev), 0.15, 4)
        zc = c['z'].to(dev).clone()
        k = 3
        idx = torch.full((2,), k, device=dev, dtype=torch.int64)
        for t in range(1, 5):
            zc = zc + 0.15 * S.forward_idx(idx, zc)
            assert rel_err(path[:, k, 4 + t], zc) < 1e-5


def test_latent_dimension_not_multiple_of_four(dev):
    """BigGAN-256 / -512 truncate dim_z to 119 / 112-ish values (BigGAN.py:106-108): the 16-byte-vector kernels run on
    zero-padded copies; forward, backward and the traversal agree with the C oracle at d = 119."""
    from oracle import wgs_oracle as O
    from tests import golden_inputs as GI
    from warpedganspace_amd.support_sets import SupportSets
    K, N, d, B = 6, 3, 119, 5
    c = GI.support_sets_case(K, N, d, B, 4119, learn_gammas=True)
    S = SupportSets(K, N, d, learn_gammas=True, gamma=c['gamma'])
    S.load_state_dict(c['sd'])
    S.to(dev)
    y = S(GI.one_hot(c['idx'], K).to(dev), c['z'].to(dev))
    (y * c['gout'].to(dev)).sum().backward()
    out, _ = O.rbf_c_forward(c['sd'], c['idx'], c['z'], True, c['gamma'])
    dtable, _, _, _ = O.rbf_c_backward(c['sd'], c['idx'], c['z'], c['gout'], True, c['gamma'])
    assert y.shape == (B, d)
    assert abs(y.detach().double().cpu().numpy() - out).max() < 1e-5
    assert abs(S.SUPPORT_SETS.grad.double().cpu().numpy() - dtable).max() / abs(dtable).max() < 1e-4
    path, shift = S.traverse(c['z'][:2].to(dev), 0.2, 3)
    ref_path, _ = O.traverse_paths(c['sd'], c['z'][:2], 0.2, 3, True, c['gamma'])
    assert path.shape == (2, K, 7, d)
    assert float((path.cpu() - ref_path).abs().max()) < 1e-4


def test_one_launch_forward_equals_the_split_forward(dev):
    """wgs_rbf_fwd is one launch (a 16-wave workgroup per sample); WGS_RBF_SPLIT restores the two-launch form (support vectors split over
    workgroups + finish).  Same field, norm and saved squared distances to fp32 summation order; the backward reads either's workspace."""
    import os
    from warpedganspace_amd import _lib as L
    lib = L.lib()
    torch.manual_seed(4)
    for K, N, d, B in ((128, 32, 512, 32), (200, 64, 512, 8), (32, 8, 128, 16), (16, 3, 1024, 5)):
        n2 = 2 * N
        table = torch.randn(K, n2 * d, device=dev) * 0.7
        alphas = torch.where(torch.arange(n2, device=dev) % 2 == 0, 1.0, -1.0).repeat(K, 1).contiguous()
        lg = (torch.randn(K, device=dev) * 0.1 - 6.0).contiguous()
        idx = torch.randint(0, K, (B,), device=dev)
        z, mag = torch.randn(B, d, device=dev), torch.rand(B, device=dev) + 0.2
        res = []
        for split in (False, True):
            if split:
                os.environ['WGS_RBF_SPLIT'] = '1'
            else:
                os.environ.pop('WGS_RBF_SPLIT', None)
            lib.wgs_dev_reload_flags()
            ws = torch.zeros(int(lib.wgs_rbf_ws_floats(B, n2, d)), device=dev)
            out = torch.empty(B, d, device=dev)
            c0 = lib.wgs_dev_launch_count()
            L.check(lib.wgs_rbf_fwd(L.ptr(table), L.ptr(alphas), L.ptr(lg), L.c_float(0.0), L.ptr(idx, torch.int64), L.ptr(z), L.ptr(mag), L.ptr(out),
                                    L.ptr(ws), B, K, n2, d, L.stream()), 'rbf')
            o_r2 = B * d + ((B + 3) & ~3)        # workspace layout (rbf.hip rbf_ws): g_raw [B,d] | gnorm [B] (padded to 4) | r2 [B,n2] | split partials
            res.append((out.clone(), torch.cat([ws[:B * d + B], ws[o_r2:o_r2 + B * n2]]).clone(), lib.wgs_dev_launch_count() - c0))
        os.environ.pop('WGS_RBF_SPLIT', None)
        lib.wgs_dev_reload_flags()
        (o1, w1, n1), (o2, w2, n2_) = res
        assert n1 == 1 and n2_ == 2
        assert (o1 - o2).abs().max() <= 2e-6 * o2.abs().max()
        assert (w1 - w2).abs().max() <= 2e-6 * w2.abs().max()
