"""LeNet branch of the Reconstructor (lib/reconstructor.py:18-49,72-75) on the HIP kernels.

Module layout = the reference's (`feature_extractor.{0,1,4,5,8,9}`, `path_indices.{0,1,3}`,
`shift_magnitudes.{0,1,3}`) so state_dicts are interchangeable.  The tiny channel counts (2c -> 6 -> 16 -> 120)
are zero-padded to multiples of 8 inside the schedule (the implicit-GEMM kernels want Ci % 8 == 0); padded
channels carry exact zeros end to end (zero weights, zero bias, zero BatchNorm gain), and their gradient
slices are dropped.
"""
import torch
from torch import nn

from . import _lib as L
from . import conv as C


def build_lenet(R):
    w = 2
    R.lenet_width = w
    R.feature_extractor = nn.Sequential(
        nn.Conv2d(R.channels * 2, 3 * w, kernel_size=(5, 5)), nn.BatchNorm2d(3 * w), nn.ReLU(), nn.MaxPool2d(kernel_size=(2, 2), stride=2),
        nn.Conv2d(3 * w, 8 * w, kernel_size=(5, 5)), nn.BatchNorm2d(8 * w), nn.ReLU(), nn.MaxPool2d(kernel_size=(2, 2), stride=2),
        nn.Conv2d(8 * w, 60 * w, kernel_size=(5, 5)), nn.BatchNorm2d(60 * w), nn.ReLU())
    R.path_indices = nn.Sequential(nn.Linear(60 * w, 42 * w), nn.BatchNorm1d(42 * w), nn.ReLU(), nn.Linear(42 * w, R.dim))
    R.shift_magnitudes = nn.Sequential(nn.Linear(60 * w, 42 * w), nn.BatchNorm1d(42 * w), nn.ReLU(), nn.Linear(42 * w, 1))


def _pad8(n):
    return (n + 7) // 8 * 8


def _packed(conv):
    w = conv.weight
    Co, Ci, kh, kw = w.shape
    if not w.permute(0, 2, 3, 1).is_contiguous():
        w.data = w.data.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return w.detach().permute(0, 2, 3, 1).reshape(Co, kh * kw, Ci)


def _padded_conv(conv, dev):
    Co, Ci = conv.weight.shape[:2]
    Cop, Cip = _pad8(Co), _pad8(Ci)
    wp = torch.zeros(Cop, 25, Cip, device=dev)
    wp[:Co, :, :Ci] = _packed(conv)
    b = torch.zeros(Cop, device=dev)
    b[:Co] = conv.bias.detach()
    return wp, b, Co, Ci, Cop, Cip


class _PadBN:
    """BatchNorm over a channel-padded tensor: gain/bias/running stats live in padded temporaries."""

    def __init__(self, bn, Cp, dev):
        self.bn, self.C, self.Cp = bn, bn.weight.shape[0], Cp
        self.g = torch.zeros(Cp, device=dev); self.g[:self.C] = bn.weight.detach()
        self.b = torch.zeros(Cp, device=dev); self.b[:self.C] = bn.bias.detach()
        self.rm = torch.zeros(Cp, device=dev); self.rm[:self.C] = bn.running_mean
        self.rv = torch.ones(Cp, device=dev); self.rv[:self.C] = bn.running_var

    def fwd(self, x, ws, train):
        N = x.numel() // self.Cp
        y = torch.empty_like(x)
        self.mean, self.invstd = torch.empty(self.Cp, device=x.device), torch.empty(self.Cp, device=x.device)
        L.check(L.lib().wgs_bn_fwd(L.ptr(x), L.ptr(self.g), L.ptr(self.b), None, L.ptr(y), L.ptr(self.mean), L.ptr(self.invstd),
                                   L.ptr(self.rm), L.ptr(self.rv), L.ptr(self.bn.num_batches_tracked, torch.int64) if train else None,
                                   L.rawptr(ws), L.c_int64(N), self.Cp, L.c_float(self.bn.eps), L.c_float(0.1), 1, int(train),
                                   L.stream()), 'lenet_bn')
        if train:
            with torch.no_grad():
                self.bn.running_mean.copy_(self.rm[:self.C])
                self.bn.running_var.copy_(self.rv[:self.C])
        return y

    def bwd(self, x, g, out, ws, train):
        N = x.numel() // self.Cp
        dx = torch.empty_like(x)
        dg, db = torch.empty(self.Cp, device=x.device), torch.empty(self.Cp, device=x.device)
        L.check(L.lib().wgs_bn_bwd(L.ptr(x), L.ptr(g), None, L.ptr(out), L.ptr(self.mean), L.ptr(self.invstd), L.ptr(self.g),
                                   L.ptr(dx), None, L.ptr(dg), L.ptr(db), L.rawptr(ws), L.c_int64(N), self.Cp, int(train),
                                   L.stream()), 'lenet_bn_bwd')
        return dx, dg[:self.C].contiguous(), db[:self.C].contiguous()


def _colsum(x, C_, ws):
    out = torch.empty(C_, device=x.device)
    L.check(L.lib().wgs_colsum(L.ptr(x), L.ptr(out), L.rawptr(ws), L.c_int64(x.numel() // C_), C_, L.stream()), 'colsum')
    return out


def _linear(x, lin):
    B, K = x.shape
    N = lin.weight.shape[0]
    y = torch.empty(B, N, device=x.device)
    L.check(L.lib().wgs_linear_fwd(L.ptr(x), L.ptr(lin.weight), L.ptr(lin.bias), L.ptr(y), B, N, K, K, N, L.c_float(1.0),
                                   L.c_float(1.0), 0, 0, L.c_float(0.0), L.c_float(1.0), L.stream()), 'lenet_linear')
    return y


def _linear_bwd(x, lin, g, grads, gbuf):
    B, K = x.shape
    N = lin.weight.shape[0]
    g = g.contiguous()
    dx = torch.empty(B, K, device=x.device)
    L.check(L.lib().wgs_linear_dgrad(L.ptr(g), L.ptr(lin.weight), None, L.ptr(dx), B, N, K, N, K, L.c_float(1.0), L.c_float(1.0),
                                     L.c_float(1.0), 0, L.stream()), 'lenet_linear_dgrad')
    dw = gbuf[id(lin.weight)] if gbuf is not None else torch.empty_like(lin.weight)
    db = gbuf[id(lin.bias)] if gbuf is not None else torch.empty_like(lin.bias)
    L.check(L.lib().wgs_linear_wgrad(L.ptr(g), L.ptr(x), L.ptr(dw), L.ptr(db), B, N, K, L.stream()), 'lenet_linear_wgrad')
    grads[id(lin.weight)], grads[id(lin.bias)] = dw, db
    return dx


def _put(grads, gbuf, p, val):
    if gbuf is not None:
        gbuf[id(p)].copy_(val.reshape(gbuf[id(p)].shape))
        grads[id(p)] = gbuf[id(p)]
    else:
        grads[id(p)] = val


def forward_impl(R, x1, x2, save=True):
    fe = R.feature_extractor
    train = R.training
    lib, st = L.lib(), L.stream()
    dev = x1.device
    x1, x2 = x1.contiguous(), x2.contiguous()
    B, c, H, W = x1.shape
    ws = torch.zeros(64 * 128, dtype=torch.float64, device=dev)      # WGS_BN_WS_DOUBLES(128)
    Cp0 = _pad8(2 * c)
    x = torch.empty(B, H, W, Cp0, device=dev)
    L.check(lib.wgs_pack_pair_nhwc(L.ptr(x1), L.ptr(x2), L.ptr(x), B, c, H * W, Cp0, st), 'pack_pair')
    stages = []
    h = x
    for ci_idx, bi_idx, pool in ((0, 1, True), (4, 5, True), (8, 9, False)):
        conv, bn = fe[ci_idx], fe[bi_idx]
        wp, b, Co, Ci, Cop, Cip = _padded_conv(conv, dev)
        cout = C.conv2d(h, wp, 5, stride=1, pad=0, bias=b, precision=0)
        pbn = _PadBN(bn, Cop, dev)
        a = pbn.fwd(cout, ws, train)
        if pool:
            Hp = a.shape[1] // 2
            pooled = torch.empty(B, Hp, a.shape[2] // 2, Cop, device=dev)
            idx = torch.empty(B, Hp, a.shape[2] // 2, Cop, dtype=torch.uint8, device=dev)
            L.check(lib.wgs_maxpool_fwd(L.ptr(a), L.ptr(pooled), L.rawptr(idx), B, a.shape[1], a.shape[2], Cop, 2, 2, 0, st), 'maxpool')
            nxt = pooled
        else:
            idx, nxt = None, a
        stages.append(dict(conv=conv, bn=bn, pbn=pbn, wp=wp, xin=h, cout=cout, a=a, idx=idx, Co=Co, Ci=Ci, Cop=Cop, Cip=Cip))
        h = nxt
    P = h.shape[1] * h.shape[2]
    Cf = h.shape[3]
    feat = torch.empty(B, Cf, device=dev)
    L.check(lib.wgs_avgpool_fwd(L.ptr(h), L.ptr(feat), B, P, Cf, st), 'mean_hw')          # features.mean([-1,-2]), :74
    heads = []
    outs = []
    for head in (R.path_indices, R.shift_magnitudes):
        t = _linear(feat, head[0])
        pbn = _PadBN(head[1], head[1].weight.shape[0], dev)
        r = pbn.fwd(t, ws, train)
        outs.append(_linear(r, head[3]))
        heads.append(dict(head=head, t=t, r=r, pbn=pbn))
    saved = dict(stages=stages, feat=feat, hshape=h.shape, heads=heads, B=B, c=c, H=H, W=W, Cp0=Cp0, train=train, ws=ws) if save else None
    mag = outs[1]
    return outs[0], (mag.reshape(B) if B > 1 else mag.squeeze()), saved


def backward_impl(R, S, dlogits, dmag, need_x=(False, True), gbuf=None):
    lib, st = L.lib(), L.stream()
    B, train, ws = S['B'], S['train'], S['ws']
    dev = dlogits.device
    grads = {}
    dfeat = None
    for hd, g in zip(S['heads'], (dlogits, dmag.reshape(B, 1))):
        head = hd['head']
        dr = _linear_bwd(hd['r'], head[3], g, grads, gbuf)
        dt, dg, db = hd['pbn'].bwd(hd['t'], dr, hd['r'], ws, train)
        _put(grads, gbuf, head[1].weight, dg)
        _put(grads, gbuf, head[1].bias, db)
        df = _linear_bwd(S['feat'], head[0], dt, grads, gbuf)
        dfeat = df if dfeat is None else dfeat + df
    hs = S['hshape']
    g = torch.empty(hs, device=dev)
    L.check(lib.wgs_avgpool_bwd(L.ptr(dfeat), L.ptr(g), B, hs[1] * hs[2], hs[3], st), 'mean_hw_bwd')
    d1 = d2 = None
    for i, sg in enumerate(reversed(S['stages'])):
        a, cout, xin = sg['a'], sg['cout'], sg['xin']
        if sg['idx'] is not None:
            da = torch.empty_like(a)
            L.check(lib.wgs_maxpool_bwd(L.ptr(g), L.rawptr(sg['idx']), L.ptr(da), B, a.shape[1], a.shape[2], sg['Cop'], 2, 2, 0, st),
                    'maxpool_bwd')
        else:
            da = g
        dc, dg, db = sg['pbn'].bwd(cout, da, a, ws, train)
        _put(grads, gbuf, sg['bn'].weight, dg)
        _put(grads, gbuf, sg['bn'].bias, db)
        _put(grads, gbuf, sg['conv'].bias, _colsum(dc, sg['Cop'], ws)[:sg['Co']].contiguous())
        dwp = torch.zeros_like(sg['wp'])
        C.conv2d_wgrad(xin, dc, dwp, 5, stride=1, pad=0)
        dw = dwp[:sg['Co'], :, :sg['Ci']].contiguous()
        if gbuf is not None:
            gbuf[id(sg['conv'].weight)].copy_(dw)
            grads[id(sg['conv'].weight)] = gbuf[id(sg['conv'].weight)]
        else:
            grads[id(sg['conv'].weight)] = dw.view(sg['Co'], 5, 5, sg['Ci']).permute(0, 3, 1, 2)
        last = (i == len(S['stages']) - 1)
        if not last or need_x[0] or need_x[1]:
            wt = C.repack_w_t(sg['wp'], sg['Cop'], 25, sg['Cip'])
            g = C.conv2d_dgrad(dc, wt, xin.shape[1:3], 5, stride=1, pad=0, precision=0)
    if need_x[0] or need_x[1]:
        c = S['c']
        d1 = torch.empty(B, c, S['H'], S['W'], device=dev) if need_x[0] else None
        d2 = torch.empty(B, c, S['H'], S['W'], device=dev) if need_x[1] else None
        L.check(lib.wgs_unpack_pair_grad(L.ptr(g), L.ptr(d1), L.ptr(d2), B, c, S['H'] * S['W'], S['Cp0'], st), 'unpack_pair')
    return grads, d1, d2
