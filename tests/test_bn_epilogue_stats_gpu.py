"""GPU: train-mode BatchNorm statistics out of the producing conv's epilogue (wgs_conv_desc.col_stats + wgs_bn_fwd_sums, round 5) — the
BatchNorm2d behind every conv of torchvision's BasicBlock (lib/reconstructor.py:52-79; oracle/wgs_oracle.py:296-319): the per-channel sums
that the separate statistics pass (wgs_bn_fwd's first launch) re-reads the tensor for, accumulated while the conv stores it."""
import pytest
import torch

from tests.util import rel_err
from warpedganspace_amd import _lib as L
from warpedganspace_amd import conv as C

pytestmark = pytest.mark.gpu

# (B, Ci, Co, H, k, stride, pad): the kernel families the Reconstructor's forward convs take at 256^2 / 1024^2 inputs — few-channel halo kernel
# (64 channels on large maps), patch kernel, register-staged tiles (strided, 1x1), split-K + second-pass epilogue (small maps)
CASES = [(4, 64, 64, 64, 3, 1, 1), (32, 128, 128, 32, 3, 1, 1), (8, 64, 128, 64, 3, 2, 1), (8, 64, 128, 64, 1, 2, 0), (32, 256, 256, 16, 3, 1, 1),
         (32, 512, 512, 8, 3, 1, 1), (2, 128, 256, 32, 3, 2, 1), (3, 64, 64, 24, 3, 1, 1), (2, 256, 512, 16, 1, 2, 0)]


def _sums(ws, Co):
    r = ws.view(32, 2, Co).sum(0)
    return r[0], r[1]


@pytest.mark.parametrize('precision', [0, 1, C.FP32W])
@pytest.mark.parametrize('B,Ci,Co,H,k,s,p', CASES)
def test_conv_epilogue_column_sums(dev, precision, B, Ci, Co, H, k, s, p):
    torch.manual_seed(Ci + Co + H + k)
    x = torch.randn(B, H, H, Ci, device=dev)
    w = torch.randn(Co, k * k, Ci, device=dev) / (k * k * Ci) ** 0.5
    ws = torch.zeros(64 * Co, dtype=torch.float64, device=dev)
    L.lib().wgs_dev_trace_kernels(1)
    y = C.conv2d(x, w, k, stride=s, pad=p, precision=precision, col_stats=ws)
    sym = L.lib().wgs_dev_last_kernel().decode()
    L.lib().wgs_dev_trace_kernels(0)
    assert torch.equal(y, C.conv2d(x, w, k, stride=s, pad=p, precision=precision)) or rel_err(y, C.conv2d(x, w, k, stride=s, pad=p, precision=precision)) < 1e-6
    s1, s2 = _sums(ws, Co)
    yd = y.double().reshape(-1, Co)
    r1, r2 = yd.sum(0), (yd * yd).sum(0)
    e1 = float((s1 - r1).abs().max() / r1.abs().max().clamp_min(1e-30))
    e2 = float((s2 - r2).abs().max() / r2.abs().max())
    print(sym, 'sum err %.1e, sum of squares err %.1e' % (e1, e2))
    assert e1 < 2e-5 and e2 < 2e-6, (sym, e1, e2)          # fp32 partial sums over <= 64 rows, fp64 from there (sum y cancels: relative to its largest entry)
    # the BatchNorm behind it: finalise + apply from the sums == the three-launch form, and the scratch is left zero
    g, b = torch.rand(Co, device=dev) + 0.5, torch.randn(Co, device=dev)
    outs = []
    for mode in ('sums', 'fused', 'full'):
        rm, rv, nb = torch.zeros(Co, device=dev), torch.ones(Co, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
        o, mean, invstd = torch.empty_like(y), torch.empty(Co, device=dev), torch.empty(Co, device=dev)
        N = y.numel() // Co
        if mode == 'fused':
            # (the conv's sums again, into a fresh buffer: the fused launch reads it, leaves it as it is and zeroes the OTHER buffer)
            ws_a = torch.zeros(64 * Co, dtype=torch.float64, device=dev)
            ws_b = torch.full((64 * Co,), 3.0, dtype=torch.float64, device=dev)
            C.conv2d(x, w, k, stride=s, pad=p, precision=precision, col_stats=ws_a)
            L.check(L.lib().wgs_bn_fwd_fused(L.ptr(y), L.ptr(g), L.ptr(b), None, L.ptr(o), L.ptr(mean), L.ptr(invstd), L.ptr(rm), L.ptr(rv),
                                             L.ptr(nb, torch.int64), L.rawptr(ws_a), L.rawptr(ws_b), L.c_int64(N), Co, L.c_float(1e-5), L.c_float(0.1), 1,
                                             L.stream()), 'bn_fused')
            nrep = max(1, min(32, 2048 // Co))
            assert float(ws_b[:nrep * 2 * Co].abs().max()) == 0.0 and float(ws_a.abs().max()) > 0.0
        elif mode == 'sums':
            L.check(L.lib().wgs_bn_fwd_sums(L.ptr(y), L.ptr(g), L.ptr(b), None, L.ptr(o), L.ptr(mean), L.ptr(invstd), L.ptr(rm), L.ptr(rv),
                                            L.ptr(nb, torch.int64), L.rawptr(ws), L.c_int64(N), Co, L.c_float(1e-5), L.c_float(0.1), 1, L.stream()), 'bn_sums')
        else:
            L.check(L.lib().wgs_bn_fwd(L.ptr(y), L.ptr(g), L.ptr(b), None, L.ptr(o), L.ptr(mean), L.ptr(invstd), L.ptr(rm), L.ptr(rv),
                                       L.ptr(nb, torch.int64), L.rawptr(ws), L.c_int64(N), Co, L.c_float(1e-5), L.c_float(0.1), 1, 1, L.stream()), 'bn')
        assert float(ws.abs().max()) == 0.0
        outs.append((o, mean, invstd, rm, rv, int(nb)))
    for other in outs[:2]:
        for a_, b_ in zip(other[:5], outs[2][:5]):
            assert rel_err(a_, b_) < 2e-5
    assert outs[0][5] == outs[1][5] == outs[2][5] == 1


def test_col_stats_argument_checks(dev):
    x = torch.randn(1, 8, 8, 24, device=dev)          # Ci % 32 != 0 in exact fp32: the plain kernel (no shared epilogue)
    w = torch.randn(64, 9, 24, device=dev)
    ws = torch.zeros(64 * 64, dtype=torch.float64, device=dev)
    with pytest.raises(L.WgsError):
        C.conv2d(x, w, 3, pad=1, precision=0, col_stats=ws)


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('arith', ['fp32', 'bf16x3', 'fp32w'])
def test_reconstructor_with_and_without_epilogue_statistics(dev, monkeypatch, arith, fused):
    """The whole Reconstructor: logits, magnitude, image gradient, every parameter gradient and the BatchNorm running statistics agree
    between the two routes (same values up to the order of fp32 / fp64 additions)."""
    from warpedganspace_amd import reconstructor as RR
    torch.manual_seed(1)
    x1, x2 = torch.randn(4, 3, 128, 128, device=dev), torch.randn(4, 3, 128, 128, device=dev)
    res = []
    for on in (True, False):
        monkeypatch.setattr(RR, 'BN_EPILOGUE_STATS', on)
        monkeypatch.setattr(RR, 'BN_FUSED_APPLY', fused)
        torch.manual_seed(2)
        R = RR.Reconstructor('ResNet', 16).to(dev).train()
        ar = RR.r_arith(arith, {'bf16x3': 1, 'fp32w': 5}.get(arith, 0))
        lib = L.lib()
        c0 = lib.wgs_dev_launch_count()
        logits, mag, saved = R._forward_impl(x1, x2, save=True, arith=ar)
        n_launch = lib.wgs_dev_launch_count() - c0
        dl, dm = torch.randn_like(logits), torch.randn(4, device=dev)
        grads, _, d2 = R._backward_impl(saved, dl, dm, need_x=(False, True))
        res.append((logits, mag, d2, [grads[id(p)].clone() for n, p in R.named_parameters() if id(p) in grads],
                    [b.clone() for b in R.buffers() if b.is_floating_point()], n_launch))
    n_bwd = None
    assert res[0][5] <= res[1][5] - (40 if fused else 20), (res[0][5], res[1][5])          # one / two launches fewer per BatchNorm forward (20 of them)
    assert rel_err(res[0][0], res[1][0]) < 2e-5 and rel_err(res[0][1], res[1][1]) < 2e-5
    for a_, b_ in zip(res[0][4], res[1][4]):          # running statistics
        assert rel_err(a_, b_) < 1e-5

    # gradients: the two routes' statistics differ in the last bits, a few of the ~1e7 ReLU gates of a batch of 4 fall on the other side, and
    # single entries move by per cent (as between ANY two fp32 evaluations of this net) — the direction is what holds
    def cos(a_, b_):
        a_, b_ = a_.double().reshape(-1), b_.double().reshape(-1)
        return float((a_ * b_).sum() / (a_.norm() * b_.norm()).clamp_min(1e-300))
    assert cos(res[0][2], res[1][2]) > 0.9999
    for a_, b_ in zip(res[0][3], res[1][3]):
        assert cos(a_, b_) > 0.999


def test_bn_backward_over_the_scratch_pair(dev):
    """wgs_bn_bwd_fused == wgs_bn_bwd (reduction, then the apply kernel sums the replicas itself), with and without the residual branch; the
    scratch it accumulated into stays dirty, the other one is left zero; a second call with the roles swapped gives the same result."""
    torch.manual_seed(3)
    for N, Cn in ((4096, 64), (700, 128), (64, 512)):
        x, dyA, dyB = torch.randn(N, Cn, device=dev), torch.randn(N, Cn, device=dev), torch.randn(N, Cn, device=dev)
        out = torch.randn(N, Cn, device=dev)
        mean, invstd = x.mean(0), 1.0 / (x.var(0, unbiased=False) + 1e-5).sqrt()
        g = torch.rand(Cn, device=dev) + 0.5
        ref = [torch.empty_like(x), torch.empty_like(x), torch.empty(Cn, device=dev), torch.empty(Cn, device=dev)]
        ws = torch.zeros(64 * Cn, dtype=torch.float64, device=dev)
        L.check(L.lib().wgs_bn_bwd(L.ptr(x), L.ptr(dyA), L.ptr(dyB), L.ptr(out), L.ptr(mean), L.ptr(invstd), L.ptr(g), L.ptr(ref[0]), L.ptr(ref[1]),
                                   L.ptr(ref[2]), L.ptr(ref[3]), L.rawptr(ws), L.c_int64(N), Cn, 1, L.stream()), 'bn_bwd')
        a, b = torch.zeros(64 * Cn, dtype=torch.float64, device=dev), torch.full((64 * Cn,), 7.0, dtype=torch.float64, device=dev)
        for _ in range(2):
            got = [torch.empty_like(x), torch.empty_like(x), torch.empty(Cn, device=dev), torch.empty(Cn, device=dev)]
            L.check(L.lib().wgs_bn_bwd_fused(L.ptr(x), L.ptr(dyA), L.ptr(dyB), L.ptr(out), L.ptr(mean), L.ptr(invstd), L.ptr(g), L.ptr(got[0]),
                                             L.ptr(got[1]), L.ptr(got[2]), L.ptr(got[3]), L.rawptr(a), L.rawptr(b), L.c_int64(N), Cn, L.stream()), 'bn_bwd_fused')
            nrep = max(1, min(32, 2048 // Cn))
            assert float(b[:nrep * 2 * Cn].abs().max()) == 0.0 and float(a.abs().max()) > 0.0
            for u, v in zip(got, ref):
                assert rel_err(u, v) < 1e-5
            a, b = b, a


def test_scratch_pair_shared_by_layers_of_different_widths(dev):
    """the fused apply launches zero a C-independent extent of the buffer they leave clean, so a producer of ANOTHER width finds it
    zero: widths 64 (32 replicas: 4096 doubles), 96 (21 replicas: 4032), 512 (4 replicas) and 2048 (one replica) in turn over one pair sized
    for the widest.  (Widths below 64 or above 2048 have their own footprint rule and must not share a pair with others: include/wgs.h.)"""
    torch.manual_seed(5)
    pair = [torch.zeros(64 * 2048, dtype=torch.float64, device=dev) for _ in range(2)]
    k = 0
    for N, Cn in ((512, 64), (300, 2048), (256, 96), (64, 512), (128, 64), (100, 2048), (700, 128)):
        x, dyA = torch.randn(N, Cn, device=dev), torch.randn(N, Cn, device=dev)
        mean, invstd = x.mean(0), 1.0 / (x.var(0, unbiased=False) + 1e-5).sqrt()
        g = torch.rand(Cn, device=dev) + 0.5
        ref = [torch.empty_like(x), torch.empty(Cn, device=dev), torch.empty(Cn, device=dev)]
        ws = torch.zeros(64 * Cn, dtype=torch.float64, device=dev)
        L.check(L.lib().wgs_bn_bwd(L.ptr(x), L.ptr(dyA), None, None, L.ptr(mean), L.ptr(invstd), L.ptr(g), L.ptr(ref[0]), None,
                                   L.ptr(ref[1]), L.ptr(ref[2]), L.rawptr(ws), L.c_int64(N), Cn, 1, L.stream()), 'bn_bwd')
        a, b = pair[k], pair[k ^ 1]
        k ^= 1
        assert float(a.abs().max()) == 0.0, (N, Cn)               # what the previous launch left for this producer
        got = [torch.empty_like(x), torch.empty(Cn, device=dev), torch.empty(Cn, device=dev)]
        L.check(L.lib().wgs_bn_bwd_fused(L.ptr(x), L.ptr(dyA), None, None, L.ptr(mean), L.ptr(invstd), L.ptr(g), L.ptr(got[0]), None,
                                         L.ptr(got[1]), L.ptr(got[2]), L.rawptr(a), L.rawptr(b), L.c_int64(N), Cn, L.stream()), 'bn_bwd_fused')
        for u, v in zip(got, ref):
            assert rel_err(u, v) < 1e-5, (N, Cn)


@pytest.mark.parametrize('precision', [0, 1])
@pytest.mark.parametrize('B,Ci,Co,H', [(32, 128, 128, 32), (8, 256, 256, 64), (32, 64, 64, 64)])
def test_column_sums_and_magnitude_bound_from_one_launch(dev, precision, B, Ci, Co, H):
    """wgs_conv_desc.col_stats and .y_amax together, alone and neither: the epilogue has one copy of its store loop per combination (the sums
    and the running maximum are computed where the value is, not behind the loop) — same y bits in all four, the right sums and maximum."""
    torch.manual_seed(B + Ci + H)
    x = torch.randn(B, H, H, Ci, device=dev)
    w = torch.randn(Co, 9, Ci, device=dev) / (9 * Ci) ** 0.5
    ref = C.conv2d(x, w, 3, pad=1, precision=precision)
    ws = torch.zeros(64 * Co, dtype=torch.float64, device=dev)
    am = torch.zeros(1, device=dev)
    y_both = C.conv2d(x, w, 3, pad=1, precision=precision, col_stats=ws, y_amax=am)
    am2 = torch.zeros(1, device=dev)
    y_am = C.conv2d(x, w, 3, pad=1, precision=precision, y_amax=am2)
    ws2 = torch.zeros(64 * Co, dtype=torch.float64, device=dev)
    y_st = C.conv2d(x, w, 3, pad=1, precision=precision, col_stats=ws2)
    assert torch.equal(y_both, ref) and torch.equal(y_am, ref) and torch.equal(y_st, ref)
    assert am.item() == am2.item() == ref.abs().max().item()
    yd = ref.double().reshape(-1, Co)
    for buf in (ws, ws2):
        s1, s2 = _sums(buf, Co)
        assert float((s1 - yd.sum(0)).abs().max() / yd.sum(0).abs().max()) < 2e-5
        assert float((s2 - (yd * yd).sum(0)).abs().max() / (yd * yd).sum(0).abs().max()) < 2e-6
