"""SupportSets — the warping network S (K RBF warping functions x N support dipoles in R^d).

Host-side mirror of the reference module (lib/support_sets.py:5-101): same constructor signature,
attributes, parameter names/shapes (`SUPPORT_SETS [K, 2N*d]`, `ALPHAS [K, 2N]`, `LOGGAMMA [K, 1]`)
and therefore the same checkpoint format.  The arithmetic runs in hand-written HIP kernels
(csrc/rbf.hip) through the C ABI (wgs_rbf_fwd / wgs_rbf_bwd / wgs_rbf_traverse); there is no
PyTorch fallback — CPU tensors raise.
"""
import math

import torch
from torch import nn

from . import _lib as L


def rbf_workspace(B, n2, d, device):
    return torch.empty(int(L.lib().wgs_rbf_ws_floats(B, n2, d)), dtype=torch.float32, device=device)


class _RbfField(torch.autograd.Function):
    """out[b] = scale[b] * g_b/|g_b| for the selected warping function idx[b] (lib/support_sets.py:81-101)."""

    @staticmethod
    def forward(ctx, table, alphas, loggamma, z, idx, scale, gamma, use_loggamma):
        B, d = z.shape
        K, n2 = alphas.shape
        z = z.contiguous()
        idx = idx.contiguous()
        out = torch.empty_like(z)
        ws = rbf_workspace(B, n2, d, z.device)
        lg = loggamma.reshape(-1) if use_loggamma else None
        L.check(L.lib().wgs_rbf_fwd(L.ptr(table, name='SUPPORT_SETS'), L.ptr(alphas, name='ALPHAS'), L.ptr(lg),
                                    L.c_float(gamma), L.ptr(idx, torch.int64, 'idx'), L.ptr(z, name='z'),
                                    L.ptr(scale), L.ptr(out), L.ptr(ws), B, K, n2, d, L.stream()), 'wgs_rbf_fwd')
        ctx.save_for_backward(table, alphas, loggamma, z, idx, scale if scale is not None else z.new_empty(0), ws)
        ctx.cfg = (gamma, use_loggamma, scale is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        table, alphas, loggamma, z, idx, scale, ws = ctx.saved_tensors
        gamma, use_loggamma, has_scale = ctx.cfg
        B, d = z.shape
        K, n2 = alphas.shape
        need_table, need_alphas, need_lg, need_z = ctx.needs_input_grad[:4]
        dtable = torch.zeros_like(table)
        dalphas = torch.zeros_like(alphas) if need_alphas else None
        dlg = torch.zeros(K, dtype=torch.float32, device=z.device) if (need_lg and use_loggamma) else None
        dz = torch.zeros_like(z) if need_z else None
        lg = loggamma.reshape(-1) if use_loggamma else None
        gout = gout.contiguous()
        L.check(L.lib().wgs_rbf_bwd(L.ptr(table), L.ptr(alphas), L.ptr(lg), L.c_float(gamma), L.ptr(idx, torch.int64),
                                    L.ptr(z), L.ptr(scale if has_scale else None), L.ptr(gout),
                                    L.ptr(ws), L.ptr(dtable), L.ptr(dlg), L.ptr(dalphas), L.ptr(dz),
                                    B, K, n2, d, L.stream()), 'wgs_rbf_bwd')
        return (dtable if need_table else None, dalphas, dlg.view_as(loggamma) if dlg is not None else None, dz,
                None, None, None, None)


class SupportSets(nn.Module):
    def __init__(self, num_support_sets, num_support_dipoles, support_vectors_dim,
                 learn_alphas=False, learn_gammas=False, gamma=None):
        """Same arguments as the reference constructor (lib/support_sets.py:6-18).

        num_support_sets (K) warping functions, each with num_support_dipoles (N) antipodal pairs of
        support vectors in R^support_vectors_dim; `gamma` defaults to 1/d as in train.py:158.
        """
        super().__init__()
        self.num_support_sets = num_support_sets
        self.num_support_dipoles = num_support_dipoles
        self.support_vectors_dim = support_vectors_dim
        self.learn_alphas = learn_alphas
        self.learn_gammas = learn_gammas
        self.gamma = (1.0 / support_vectors_dim) if gamma is None else gamma
        self.loggamma = torch.log(torch.scalar_tensor(self.gamma))
        K, N, d = num_support_sets, num_support_dipoles, support_vectors_dim

        # Init law of lib/support_sets.py:39-54: radii arange(1, 4, 3/K); per set N random unit
        # directions, each paired with its antipode, scaled to the set's radius.
        self.r_min, self.r_max = 1.0, 4.0
        self.radii = torch.arange(self.r_min, self.r_max, (self.r_max - self.r_min) / K)[:K]
        v = torch.randn(K, N, d)
        v = v / v.norm(dim=2, keepdim=True)
        sv = torch.stack((v, -v), dim=2).reshape(K, 2 * N, d) * self.radii.view(K, 1, 1).to(v)
        self.SUPPORT_SETS = nn.Parameter(sv.reshape(K, 2 * N * d).contiguous(), requires_grad=True)
        # alphas +1,-1,+1,-1,...  (:63-70); LOGGAMMA = log(gamma) (:78-79)
        self.ALPHAS = nn.Parameter(torch.tensor([1.0, -1.0]).repeat(N).unsqueeze(0).repeat(K, 1).contiguous(),
                                   requires_grad=learn_alphas)
        self.LOGGAMMA = nn.Parameter(self.loggamma * torch.ones(K, 1), requires_grad=learn_gammas)

    # -- fast path -------------------------------------------------------------------------------
    def forward_idx(self, idx, z, scale=None):
        """Unit-norm gradient field of warping function idx[b] at z[b]; optional per-sample `scale`
        (the trainer's shift magnitude, lib/trainer.py:235) is fused into the kernel."""
        d = self.support_vectors_dim
        pad = (-d) % 4
        if pad:     # the kernels stream 16-byte vectors: zero-pad the latent dimension (BigGAN-256: 119); autograd slices back
            K, n2 = self.ALPHAS.shape
            table = torch.nn.functional.pad(self.SUPPORT_SETS.view(K, n2, d), (0, pad)).reshape(K, -1)
            out = _RbfField.apply(table, self.ALPHAS, self.LOGGAMMA, torch.nn.functional.pad(z, (0, pad)), idx, scale,
                                  float(self.gamma), bool(self.learn_gammas))
            return out[:, :d]
        return _RbfField.apply(self.SUPPORT_SETS, self.ALPHAS, self.LOGGAMMA, z, idx, scale, float(self.gamma),
                               bool(self.learn_gammas))

    # -- reference signature -----------------------------------------------------------------------
    def forward(self, support_sets_mask, z):
        """`support_sets_mask` [B,K] is the reference's one-hot selector (lib/trainer.py:227-231); the selected path index
        is its argmax.  The reference gathers the selected rows with `mask @ SUPPORT_SETS` (lib/support_sets.py:84-88), which
        also accepts soft / multi-hot masks (a blend of warping functions): the kernel evaluates ONE function per sample, so
        anything but a strict one-hot row is rejected here instead of silently picking one path (this check syncs; the
        training loop uses forward_idx)."""
        m = support_sets_mask
        if m.dim() != 2 or m.shape[1] != self.num_support_sets or not bool(((m.sum(1) == 1) & (m.max(1).values == 1)).all()):
            raise L.WgsError("SupportSets.forward needs a strict one-hot mask [B, K] (one warping function per sample); "
                             "soft / multi-hot masks are not supported on the HIP path")
        return self.forward_idx(torch.argmax(m, dim=1), z)

    @torch.no_grad()
    def traverse(self, codes, eps, steps):
        """All-K walks of traverse_latent_space.py:361-438 in one launch.
        Returns (path [n,K,2*steps+1,d], shift [n,K,2*steps+1,d])."""
        n, d0 = codes.shape
        K, n2 = self.ALPHAS.shape
        pad = (-d0) % 4
        d = d0 + pad
        table = self.SUPPORT_SETS
        if pad:     # 16-byte vectors in the kernel: zero-pad the latent dimension (see forward_idx)
            table = torch.nn.functional.pad(table.view(K, n2, d0), (0, pad)).reshape(K, -1).contiguous()
            codes = torch.nn.functional.pad(codes, (0, pad))
        path = torch.empty(n, K, 2 * steps + 1, d, device=codes.device)
        shift = torch.empty_like(path)
        lg = self.LOGGAMMA.reshape(-1) if self.learn_gammas else None
        codes = codes.contiguous()
        L.check(L.lib().wgs_rbf_traverse(L.ptr(table), L.ptr(self.ALPHAS), L.ptr(lg),
                                         L.c_float(float(self.gamma)), L.ptr(codes), L.c_float(eps),
                                         steps, L.ptr(path), L.ptr(shift), n, K, n2, d, L.stream()),
                'wgs_rbf_traverse')
        if pad:
            path, shift = path[..., :d0].contiguous(), shift[..., :d0].contiguous()
        return path, shift
