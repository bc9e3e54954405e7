"""wgs_oracle.py — TEST INFRASTRUCTURE (oracle).

Plain PyTorch-CPU fp32 restatements of the reference algorithms on the hot path.  Functional style:
every function takes the weights as a dict keyed exactly like the reference's `state_dict()` so the
same tensors can be fed to the reference modules (when generating golden vectors in the build
container, tools/make_golden.py), to this oracle, and to the HIP product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Citations are file:line under the reference tree.
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))


# =================================================================================================
# RBF warping field — lib/support_sets.py:81-101
# =================================================================================================
def support_sets_forward(sd, mask, z, learn_gammas, gamma):
    """sd: {'SUPPORT_SETS','ALPHAS','LOGGAMMA'}; mask [B,K] one-hot; z [B,d] -> unit field [B,d]."""
    K, two_n_d = sd['SUPPORT_SETS'].shape
    n2 = sd['ALPHAS'].shape[1]
    d = two_n_d // n2
    sv = (mask @ sd['SUPPORT_SETS']).reshape(-1, n2, d)              # :83-84
    alphas = (mask @ sd['ALPHAS']).unsqueeze(2)                       # :87
    if learn_gammas:
        gammas = torch.exp(mask @ sd['LOGGAMMA']).unsqueeze(2)        # :90-91
    else:
        gammas = gamma * torch.ones(z.shape[0], n2, 1)                # :93
    D = z.unsqueeze(1).repeat(1, n2, 1) - sv                          # :96
    r2 = (torch.norm(D, dim=2) ** 2).unsqueeze(2)
    grad_f = -2 * (alphas * gammas * torch.exp(-gammas * r2) * D).sum(dim=1)   # :97-98
    return grad_f / torch.norm(grad_f, dim=1, keepdim=True)          # :101


def support_sets_init(K, N, d, gamma, generator=None):
    """State dict with the reference's initialisation law (lib/support_sets.py:35-79): radii
    arange(1,4,3/K), N random antipodal pairs per set, alphas +1/-1, loggamma = log(gamma)."""
    radii = torch.arange(1.0, 4.0, 3.0 / K)[:K]
    v = torch.randn(K, N, d, generator=generator)
    v = v / v.norm(dim=2, keepdim=True)
    sv = torch.stack([v, -v], dim=2).reshape(K, 2 * N, d) * radii.view(K, 1, 1)
    alphas = torch.tensor([1.0, -1.0]).repeat(N).unsqueeze(0).repeat(K, 1)
    return {'SUPPORT_SETS': sv.reshape(K, 2 * N * d).contiguous(), 'ALPHAS': alphas.contiguous(),
            'LOGGAMMA': math.log(gamma) * torch.ones(K, 1)}


_rbf_c = None


def rbf_c_lib():
    """The plain-C fp64 restatement (oracle/rbf_ref.c), built by __graft_entry__.build()."""
    global _rbf_c
    if _rbf_c is None:
        path = os.path.join(_HERE, '_build', 'librbf_ref.so')
        if not os.path.isfile(path):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            import subprocess
            subprocess.check_call(['gcc', '-O2', '-shared', '-fPIC', os.path.join(_HERE, 'rbf_ref.c'),
                                   '-o', path, '-lm'])
        _rbf_c = ctypes.CDLL(path)
    return _rbf_c


def _np32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


def rbf_c_forward(sd, idx, z, learn_gammas, gamma):
    lib = rbf_c_lib()
    table, alphas = _np32(sd['SUPPORT_SETS']), _np32(sd['ALPHAS'])
    lg = _np32(sd['LOGGAMMA'].reshape(-1)) if learn_gammas else None
    idx = np.ascontiguousarray(idx.cpu().numpy().astype(np.int64))
    zz = _np32(z)
    B, d = zz.shape
    K, n2 = alphas.shape
    out = np.zeros((B, d), np.float64)
    graw = np.zeros((B, d), np.float64)
    P = ctypes.c_void_p
    lib.rbf_ref_forward(P(table.ctypes.data), P(alphas.ctypes.data), P(lg.ctypes.data if lg is not None else 0),
                        ctypes.c_double(gamma), P(idx.ctypes.data), P(zz.ctypes.data), P(out.ctypes.data),
                        P(graw.ctypes.data), B, K, n2, d)
    return out, graw


def rbf_c_backward(sd, idx, z, gout, learn_gammas, gamma):
    lib = rbf_c_lib()
    table, alphas = _np32(sd['SUPPORT_SETS']), _np32(sd['ALPHAS'])
    lg = _np32(sd['LOGGAMMA'].reshape(-1)) if learn_gammas else None
    idx = np.ascontiguousarray(idx.cpu().numpy().astype(np.int64))
    zz, go = _np32(z), _np32(gout)
    B, d = zz.shape
    K, n2 = alphas.shape
    dtable = np.zeros((K, n2 * d), np.float64)
    dal = np.zeros((K, n2), np.float64)
    dlg = np.zeros((K,), np.float64)
    dz = np.zeros((B, d), np.float64)
    P = ctypes.c_void_p
    lib.rbf_ref_backward(P(table.ctypes.data), P(alphas.ctypes.data), P(lg.ctypes.data if lg is not None else 0),
                         ctypes.c_double(gamma), P(idx.ctypes.data), P(zz.ctypes.data), P(go.ctypes.data),
                         P(dtable.ctypes.data), P(dal.ctypes.data), P(dlg.ctypes.data), P(dz.ctypes.data),
                         B, K, n2, d)
    return dtable, dal, dlg, dz


def traverse_paths(sd, codes, eps, T, learn_gammas, gamma):
    """All-K latent walks, traverse_latent_space.py:361-438 with shift_leap = 1:
    returns (path [n,K,2T+1,d], shift [n,K,2T+1,d])."""
    K = sd['ALPHAS'].shape[0]
    n, d = codes.shape
    path = torch.zeros(n, K, 2 * T + 1, d)
    shift = torch.zeros(n, K, 2 * T + 1, d)
    for c in range(n):
        for k in range(K):
            mask = torch.zeros(1, K)
            mask[0, k] = 1.0
            path[c, k, T] = codes[c]
            for sign, step_idx in ((1.0, lambda t: T + t), (-1.0, lambda t: T - t)):
                zc = codes[c:c + 1].clone()
                for t in range(1, T + 1):
                    sh = sign * eps * support_sets_forward(sd, mask, zc, learn_gammas, gamma)
                    zc = zc + sh
                    path[c, k, step_idx(t)] = zc[0]
                    shift[c, k, step_idx(t)] = sh[0]
    return path, shift


# =================================================================================================
# StyleGAN2 native ops — models/StyleGAN2/op/*
# =================================================================================================
def fused_bias_act(x, bias=None, ref=None, act=3, grad=0, alpha=0.2, scale=2 ** 0.5):
    """fused_bias_act_kernel.cu:25-47: y = act(x + b[channel dim 1]) * scale."""
    if bias is not None and bias.numel():
        x = x + bias.view(1, -1, *([1] * (x.ndim - 2)))
    mode = act * 10 + grad
    if mode in (10, 11):
        y = x
    elif mode in (12, 32):
        y = torch.zeros_like(x)
    elif mode == 30:
        y = torch.where(x > 0, x, x * alpha)
    elif mode == 31:
        y = torch.where(ref > 0, x, x * alpha)
    else:
        y = x
    return y * scale


# Test hook: when set to an iterator of boolean masks, successive fused_leaky_relu calls take their gate
# (v > 0) from it instead of from their own pre-activation.  Lets a test differentiate the oracle through
# exactly the piecewise-linear branch another fp32 evaluation took (a few of ~1e7 pre-activations round
# to the other side of zero between any two implementations).
GATE_OVERRIDE = None


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    """FusedLeakyReLUFunction.forward, op/fused_act.py:51-59 (autograd of these torch ops is the
    same function the reference's hand-written backward computes, :19-48)."""
    v = x + bias.view(1, -1, *([1] * (x.ndim - 2)))
    if GATE_OVERRIDE is not None:
        gate = next(GATE_OVERRIDE).to(v.device)
        return torch.where(gate, v, v * negative_slope) * scale
    return F.leaky_relu(v, negative_slope) * scale


def upfirdn2d_mhwc(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """Semantics of upfirdn2d_native (op/upfirdn2d.py:152-186) on the op's [major, h, w, minor]
    layout, written from its definition: zero-insert upsample, pad (negative pad = crop), correlate
    every (major, minor) plane with the FLIPPED kernel, keep every down-th sample."""
    major, in_h, in_w, minor = x.shape
    kh, kw = kernel.shape
    planes = x.permute(0, 3, 1, 2).reshape(major * minor, 1, in_h, in_w)
    up = planes.new_zeros(major * minor, 1, in_h * up_y, in_w * up_x)
    up[:, :, ::up_y, ::up_x] = planes                       # sample (iy,ix) sits at (iy*up_y, ix*up_x)
    up = F.pad(up, [pad_x0, pad_x1, pad_y0, pad_y1])        # F.pad crops for negative amounts
    full = F.conv2d(up, torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw))
    full = full[:, :, ::down_y, ::down_x]
    oh, ow = full.shape[2], full.shape[3]
    return full.reshape(major, minor, oh, ow).permute(0, 2, 3, 1).contiguous()


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """upfirdn2d(), op/upfirdn2d.py:144-149, on NCHW input."""
    b, c, h, w = x.shape
    out = upfirdn2d_mhwc(x.reshape(-1, h, w, 1), kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
    return out.reshape(b, c, out.shape[1], out.shape[2])


def make_blur_kernel(k=(1, 3, 3, 1)):
    """make_kernel, models/StyleGAN2/model.py:18-26."""
    k = torch.tensor(k, dtype=torch.float32)
    k = k[None, :] * k[:, None]
    return k / k.sum()


# =================================================================================================
# StyleGAN2 generator — models/StyleGAN2/model.py, functional over the reference state_dict
# =================================================================================================
def sg2_pixel_norm(x):
    """PixelNorm.forward, model.py:14-15."""
    return x * torch.rsqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)


def sg2_equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    """EqualLinear.forward, model.py:126-131."""
    scale = (1 / math.sqrt(weight.shape[1])) * lr_mul
    if activation:
        return fused_leaky_relu(F.linear(x, weight * scale), bias * lr_mul)
    return F.linear(x, weight * scale, bias=bias * lr_mul)


def sg2_mapping(sd, z, n_mlp=8, lr_mlp=0.01):
    """Generator.style (model.py:290-295): PixelNorm then n_mlp activated EqualLinears."""
    x = sg2_pixel_norm(z)
    for i in range(1, n_mlp + 1):
        x = sg2_equal_linear(x, sd['style.%d.weight' % i], sd['style.%d.bias' % i], lr_mul=lr_mlp, activation=True)
    return x


def sg2_modulated_conv(sd, prefix, x, style, demodulate=True, upsample=False):
    """ModulatedConv2d.forward, model.py:187-228 — as written in the reference: per-sample weights
    are materialised and applied with a grouped convolution (groups = batch)."""
    weight = sd[prefix + '.weight']                       # [1, Co, Ci, k, k]
    _, co, ci, k, _ = weight.shape
    b, _, h, w = x.shape
    s = sg2_equal_linear(style, sd[prefix + '.modulation.weight'], sd[prefix + '.modulation.bias']).view(b, 1, ci, 1, 1)
    wgt = (1 / math.sqrt(ci * k * k)) * weight * s        # :191
    if demodulate:
        demod = torch.rsqrt(wgt.pow(2).sum([2, 3, 4]) + 1e-8)      # :194
        wgt = wgt * demod.view(b, co, 1, 1, 1)
    if upsample:
        xin = x.reshape(1, b * ci, h, w)
        wt = wgt.transpose(1, 2).reshape(b * ci, co, k, k)
        out = F.conv_transpose2d(xin, wt, padding=0, stride=2, groups=b)       # :209
        out = out.view(b, co, out.shape[2], out.shape[3])
        return upfirdn2d(out, sd[prefix + '.blur.kernel'], pad=(1, 1))        # Blur pad: p=(4-2)-(3-1)=0 -> (1,1), :160-165
    xin = x.reshape(1, b * ci, h, w)
    out = F.conv2d(xin, wgt.view(b * co, ci, k, k), padding=k // 2, groups=b)  # :224
    return out.view(b, co, out.shape[2], out.shape[3])


def sg2_styled_conv(sd, prefix, x, style, noise, upsample=False):
    """StyledConv.forward, model.py:266-267: activate(noise(conv(x, style)))."""
    out = sg2_modulated_conv(sd, prefix + '.conv', x, style, upsample=upsample)
    out = out + sd[prefix + '.noise.weight'] * noise                             # NoiseInjection :236-241
    return fused_leaky_relu(out, sd[prefix + '.activate.bias'])


def sg2_to_rgb(sd, prefix, x, style, skip=None):
    """ToRGB.forward, model.py:278-282."""
    out = sg2_modulated_conv(sd, prefix + '.conv', x, style, demodulate=False) + sd[prefix + '.bias']
    if skip is not None:
        out = out + upfirdn2d(skip, sd[prefix + '.upsample.kernel'], up=2, pad=(2, 1))
    return out


def sg2_synthesis(sd, w, size):
    """Generator.forward with input_is_latent=True, one style, registered noise (model.py:364-403)."""
    log_size = int(math.log(size, 2))
    b = w.shape[0]
    out = sd['input.input'].repeat(b, 1, 1, 1)
    out = sg2_styled_conv(sd, 'conv1', out, w, sd['noises.noise_0'])
    skip = sg2_to_rgb(sd, 'to_rgb1', out, w)
    for j in range(log_size - 2):
        out = sg2_styled_conv(sd, 'convs.%d' % (2 * j), out, w, sd['noises.noise_%d' % (2 * j + 1)], upsample=True)
        out = sg2_styled_conv(sd, 'convs.%d' % (2 * j + 1), out, w, sd['noises.noise_%d' % (2 * j + 2)])
        skip = sg2_to_rgb(sd, 'to_rgbs.%d' % j, out, w, skip)
    return skip


def sg2_generate(sd, z, size, shift=None, shift_in_w_space=False):
    """StyleGAN2Wrapper.forward, models/gan_load.py:157-179."""
    if shift_in_w_space:
        w = sg2_mapping(sd, z)
        return sg2_synthesis(sd, w if shift is None else w + shift, size)
    return sg2_synthesis(sd, sg2_mapping(sd, z if shift is None else z + shift), size)


# =================================================================================================
# Reconstructor — lib/reconstructor.py:10-79, functional over the reference state_dict
# =================================================================================================
def _bn(sd, prefix, x, training=True):
    """nn.BatchNorm2d/1d forward; updates the running statistics in `sd` in place like the module does."""
    rm, rv = sd[prefix + '.running_mean'], sd[prefix + '.running_var']
    if training and (prefix + '.num_batches_tracked') in sd:
        sd[prefix + '.num_batches_tracked'] += 1
    return F.batch_norm(x, rm, rv, sd[prefix + '.weight'], sd[prefix + '.bias'], training=training, momentum=0.1, eps=1e-5)


def resnet18_features(sd, x, prefix='features_extractor', training=True):
    """torchvision ResNet-18 (BasicBlock x [2,2,2,2]) up to and including the global average pool, the
    tensor the reference captures with its avgpool forward hook (lib/reconstructor.py:6-7,62-63,78).
    The arithmetic lives in the un-vendored torchvision dependency; this follows its public definition:
    conv7x7/2(pad 3, no bias) - BN - ReLU - maxpool3/2(pad 1) - stages (64,128,256,512), stride 2 from the
    second stage with a 1x1/2 conv + BN shortcut - AdaptiveAvgPool2d(1)."""
    p = prefix + '.'
    h = F.conv2d(x, sd[p + 'conv1.weight'], stride=2, padding=3)
    h = F.relu(_bn(sd, p + 'bn1', h, training))
    h = F.max_pool2d(h, 3, 2, 1)
    for li in range(1, 5):
        for bi in range(2):
            q = '%slayer%d.%d.' % (p, li, bi)
            stride = 2 if (li > 1 and bi == 0) else 1
            out = F.conv2d(h, sd[q + 'conv1.weight'], stride=stride, padding=1)
            out = F.relu(_bn(sd, q + 'bn1', out, training))
            out = F.conv2d(out, sd[q + 'conv2.weight'], stride=1, padding=1)
            out = _bn(sd, q + 'bn2', out, training)
            if (q + 'downsample.0.weight') in sd:
                ident = _bn(sd, q + 'downsample.1', F.conv2d(h, sd[q + 'downsample.0.weight'], stride=stride), training)
            else:
                ident = h
            h = F.relu(out + ident)
    return F.adaptive_avg_pool2d(h, 1).flatten(1)


def reconstructor_resnet(sd, x1, x2, training=True):
    """Reconstructor.forward, ResNet branch (lib/reconstructor.py:76-79). The reference also evaluates
    torchvision's fc layer and discards the result; it is omitted here (no observable effect)."""
    feat = resnet18_features(sd, torch.cat([x1, x2], dim=1), training=training)
    logits = F.linear(feat, sd['path_indices.weight'], sd['path_indices.bias'])
    mag = F.linear(feat, sd['shift_magnitudes.weight'], sd['shift_magnitudes.bias']).squeeze()
    return logits, mag


def reconstructor_lenet(sd, x1, x2, training=True):
    """Reconstructor.forward, LeNet branch (lib/reconstructor.py:18-49,72-75)."""
    h = torch.cat([x1, x2], dim=1)
    for i in (0, 4, 8):
        h = F.conv2d(h, sd['feature_extractor.%d.weight' % i], sd['feature_extractor.%d.bias' % i])
        h = F.relu(_bn(sd, 'feature_extractor.%d' % (i + 1), h, training))
        if i != 8:
            h = F.max_pool2d(h, 2, 2)
    feat = h.mean(dim=[-1, -2]).view(x1.shape[0], -1)
    outs = []
    for head in ('path_indices', 'shift_magnitudes'):
        t = F.linear(feat, sd[head + '.0.weight'], sd[head + '.0.bias'])
        t = F.relu(_bn(sd, head + '.1', t, training))
        outs.append(F.linear(t, sd[head + '.3.weight'], sd[head + '.3.bias']))
    return outs[0], outs[1].squeeze()


def training_loss(logits, mag_pred, target_idx, target_mag, lambda_cls=1.0, lambda_reg=0.25):
    """lib/trainer.py:245-249 (+ accuracy :257-258)."""
    ce = F.cross_entropy(logits, target_idx)
    l1 = torch.mean(torch.abs(mag_pred - target_mag))
    acc = torch.mean((torch.argmax(logits, dim=1) == target_idx).to(torch.float32))
    return lambda_cls * ce + lambda_reg * l1, ce, l1, acc


def adam_step(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (lib/trainer.py:153-156), single tensor, in place."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    p.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)


# =================================================================================================
# One training step as written in the reference — lib/trainer.py:190-254
# =================================================================================================
class ReferenceStep:
    """Holds leaf copies of the G / S / R state dicts and replays the reference's loop body with plain
    torch autograd: G(z) WITH a graph, warp, G(z + shift), R, CE + lambda*L1, loss.backward() (which also
    builds the generator's never-used weight gradients, exactly like the reference), two Adam steps."""

    def __init__(self, sd_g, sd_s, sd_r, size, learn_gammas=True, gamma=None, lambda_cls=1.0, lambda_reg=0.25,
                 lr_s=1e-4, lr_r=1e-4, shift_in_w_space=False, g_requires_grad=True, reconstructor='ResNet',
                 generator='StyleGAN2', gen_kwargs=None):
        """generator: 'StyleGAN2' (size = resolution), 'SNGAN' (gen_kwargs: channels=...), 'ProgGAN' (num_blocks=...),
        'BigGAN' (gen_kwargs: class_ids=..., resolution=...)."""
        self.size, self.learn_gammas, self.gamma = size, learn_gammas, gamma
        self.lc, self.lr_, self.w_space, self.rtype = lambda_cls, lambda_reg, shift_in_w_space, reconstructor
        self.gtype, self.gkw = generator, (gen_kwargs or {})
        self.g = {k: v.detach().clone().requires_grad_(g_requires_grad and v.is_floating_point() and
                                                        not k.startswith('noises.') and not k.endswith('kernel') and
                                                        'running' not in k and 'stored' not in k and not k.endswith('.u0') and
                                                        not k.endswith('.sv0'))
                  for k, v in sd_g.items()}
        self.s = {k: v.detach().clone() for k, v in sd_s.items()}
        self.s['SUPPORT_SETS'].requires_grad_(True)
        if learn_gammas:
            self.s['LOGGAMMA'].requires_grad_(True)
        self.r = {}
        for k, v in sd_r.items():
            v = v.detach().clone().contiguous()
            trainable = v.is_floating_point() and not (k.endswith('running_mean') or k.endswith('running_var'))
            self.r[k] = v.requires_grad_(True) if trainable else v
        self.opt = {'s': {}, 'r': {}}
        self.t = 0
        self.lrs = {'s': lr_s, 'r': lr_r}

    def _gen(self, z, shift=None):
        if self.gtype == 'StyleGAN2':
            return sg2_generate(self.g, z, self.size, shift, shift_in_w_space=self.w_space)
        if self.gtype == 'SNGAN':
            return sngan_generate(self.g, z, shift, **self.gkw)
        if self.gtype == 'ProgGAN':
            return proggan_generate(self.g, z, shift, **self.gkw)
        if self.gtype == 'BigGAN':      # gen_kwargs: class_ids (fixed per sample here; the wrapper draws them), resolution
            kw = dict(self.gkw)
            return biggan_generate(self.g, z, kw.pop('class_ids'), shift, **kw)
        raise ValueError(self.gtype)

    def step(self, z, idx, mag):
        K = self.s['ALPHAS'].shape[0]
        for d in (self.g, self.s, self.r):
            for v in d.values():
                if v.is_floating_point() and v.requires_grad:
                    v.grad = None
        img = self._gen(z)                                                                    # :200
        mask = torch.zeros(z.shape[0], K)
        mask[torch.arange(z.shape[0]), idx] = 1.0                                            # :227-231
        code = sg2_mapping(self.g, z) if self.w_space else z
        shift = mag.reshape(-1, 1) * support_sets_forward(self.s, mask, code, self.learn_gammas, self.gamma)   # :235
        img_shifted = self._gen(z, shift)                                                     # :239
        fn = reconstructor_resnet if self.rtype == 'ResNet' else reconstructor_lenet
        logits, mag_hat = fn(self.r, img, img_shifted, training=True)                        # :242
        loss, ce, l1, acc = training_loss(logits, mag_hat, idx, mag, self.lc, self.lr_)      # :245-249
        loss.backward()                                                                      # :250
        self.t += 1
        for grp, d in (('s', self.s), ('r', self.r)):                                        # :253-254
            for k, v in d.items():
                if v.is_floating_point() and v.requires_grad and v.grad is not None:
                    stt = self.opt[grp].setdefault(k, (torch.zeros_like(v), torch.zeros_like(v)))
                    with torch.no_grad():
                        adam_step(v, v.grad, stt[0], stt[1], self.t, lr=self.lrs[grp])
        return dict(loss=loss.item(), ce=ce.item(), l1=l1.item(), acc=acc.item(), argmax=torch.argmax(logits, 1),
                    logits=logits.detach(), img=img.detach(), img_shifted=img_shifted.detach(), shift=shift.detach())


# =================================================================================================
# ProgGAN generator — models/ProgGAN/model.py:12-95, functional over the reference state_dict
# =================================================================================================
PROGGAN_UP = [False, False, True, False, True, False, True, False, True, False, True, False, True, False, True, False, True, False]
PROGGAN_PAD = [3] + [1] * 17


def _leaky_relu(v, slope):
    """F.leaky_relu, or the GATE_OVERRIDE test hook (see fused_leaky_relu)."""
    if GATE_OVERRIDE is not None:
        return torch.where(next(GATE_OVERRIDE).to(v.device), v, v * slope)
    return F.leaky_relu(v, negative_slope=slope)


def proggan_pixel_norm(x):
    """PixelNormLayer.forward, model.py:17-18."""
    return x / torch.sqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)


def proggan_generate(sd, z, shift=None, num_blocks=18):
    """ProgGANWrapper.forward (models/gan_load.py:115-120) + Generator.forward (model.py:92-95)."""
    x = (z if shift is None else z + shift).reshape(z.shape[0], z.shape[1], 1, 1)
    for i in range(num_blocks):
        x = proggan_pixel_norm(x)                                                   # NormConvBlock / NormUpscaleConvBlock
        if PROGGAN_UP[i]:
            x = F.interpolate(x, scale_factor=2, mode='nearest')                    # :53
        x = F.conv2d(x, sd['features.%d.conv.weight' % i], padding=PROGGAN_PAD[i])
        x = x * sd['features.%d.wscale.scale' % i] + sd['features.%d.wscale.b' % i].view(1, -1, 1, 1)   # WScaleLayer :28-32
        x = _leaky_relu(x, 0.2)
    x = proggan_pixel_norm(x)
    x = F.conv2d(x, sd['output.conv.weight'])
    return x * sd['output.wscale.scale'] + sd['output.wscale.b'].view(1, -1, 1, 1)


# =================================================================================================
# SNGAN ResNet generator — models/SNGAN/sn_gen_resnet.py:24-112 (eval mode), over the GenWrapper state_dict
# =================================================================================================
def _relu(v):
    if GATE_OVERRIDE is not None:
        return torch.where(next(GATE_OVERRIDE).to(v.device), v, torch.zeros_like(v))
    return F.relu(v)


def sngan_generate(sd, z, shift=None, channels=(256, 256, 256, 256), seed_dim=4):
    """SNGANWrapper.forward (models/gan_load.py:27-28) on make_resnet_generator's Sequential (:81-112)."""
    def bn(prefix, x):
        return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'], sd[prefix + '.weight'],
                            sd[prefix + '.bias'], training=False, eps=1e-5)
    x = z if shift is None else z + shift
    x = F.linear(x, sd['model.0.weight'], sd['model.0.bias']).view(-1, channels[0], seed_dim, seed_dim)
    for i in range(len(channels) - 1):
        p = 'model.%d.' % (2 + i)
        h = _relu(bn(p + 'model.0', x))
        h = F.interpolate(h, scale_factor=2)
        h = F.conv2d(h, sd[p + 'conv1.weight'], sd[p + 'conv1.bias'], padding=1)
        h = _relu(bn(p + 'model.4', h))
        h = F.conv2d(h, sd[p + 'conv2.weight'], sd[p + 'conv2.bias'], padding=1)
        byp = F.interpolate(x, scale_factor=2)
        if (p + 'bypass.1.weight') in sd:
            byp = F.conv2d(byp, sd[p + 'bypass.1.weight'], sd[p + 'bypass.1.bias'], padding=1)
        x = h + byp                                                            # ResBlockGenerator.forward :53-54
    n = 2 + len(channels) - 1
    x = _relu(bn('model.%d' % n, x))
    x = F.conv2d(x, sd['model.%d.weight' % (n + 2)], sd['model.%d.bias' % (n + 2)], padding=1)
    return torch.tanh(x)


# =================================================================================================
# BigGAN generator — models/BigGAN/BigGAN.py:222-243 + layers.py, eval mode, over the reference state_dict
# =================================================================================================
def _sn_weight(sd, prefix, eps):
    """SN.W_ with update=False (layers.py:84-96) using ONE power-iteration step on the stored u0 (:24-47)."""
    W = sd[prefix + '.weight']
    Wm = W.reshape(W.shape[0], -1)
    with torch.no_grad():
        v = F.normalize(sd[prefix + '.u0'] @ Wm, eps=eps)
        u = F.normalize(v @ Wm.t(), eps=eps)
    sv = torch.squeeze((v @ Wm.t()) @ u.t())
    return W / sv


def biggan_generate(sd, z, class_ids, shift=None, ch=96, resolution=128, shared_dim=128, attention_res=64, bn_eps=1e-5,
                    sn_eps=1e-6, bottom_width=4):
    """BigGANWrapper.forward (models/gan_load.py:79-81) + Generator.forward with hier=True, G_shared=True."""
    # channel plans of models/BigGAN/BigGAN.py:23-36 (in multiples of ch, out multiples, block output resolutions)
    arch = {128: ([16, 16, 8, 4, 2], [16, 8, 4, 2, 1], [8, 16, 32, 64, 128]),
            256: ([16, 16, 8, 8, 4, 2], [16, 8, 8, 4, 2, 1], [8, 16, 32, 64, 128, 256])}[resolution]
    z = z if shift is None else z + shift
    y = F.embedding(class_ids, sd['shared.weight'])
    nslots = len(arch[0]) + 1
    cs = z.shape[1] // nslots           # BigGAN.py:106-108: dim_z is truncated to a multiple of the slot count (120 -> 119 at 256)
    zs = torch.split(z[:, :cs * nslots], cs, 1)
    ys = [torch.cat([y, item], 1) for item in zs[1:]]
    h = F.linear(zs[0], _sn_weight(sd, 'linear', sn_eps), sd['linear.bias'])
    h = h.view(h.size(0), -1, bottom_width, bottom_width)

    def ccbn(prefix, x, yb):
        gain = (1 + F.linear(yb, _sn_weight(sd, prefix + '.gain', sn_eps))).view(yb.size(0), -1, 1, 1)
        bias = F.linear(yb, _sn_weight(sd, prefix + '.bias', sn_eps)).view(yb.size(0), -1, 1, 1)
        out = F.batch_norm(x, sd[prefix + '.stored_mean'], sd[prefix + '.stored_var'], None, None, False, 0.1, bn_eps)
        return out * gain + bias

    def conv(prefix, x, pad):
        return F.conv2d(x, _sn_weight(sd, prefix, sn_eps), sd.get(prefix + '.bias'), padding=pad)

    for i, res in enumerate(arch[2]):
        p = 'blocks.%d.0.' % i
        x = h
        t = _relu(ccbn(p + 'bn1', x, ys[i]))                               # GBlock.forward, layers.py:393-405
        t = F.interpolate(t, scale_factor=2)
        x = F.interpolate(x, scale_factor=2)
        t = conv(p + 'conv1', t, 1)
        t = _relu(ccbn(p + 'bn2', t, ys[i]))
        t = conv(p + 'conv2', t, 1)
        h = t + conv(p + 'conv_sc', x, 0)
        if res == attention_res:                                           # Attention.forward, layers.py:153-166
            a = 'blocks.%d.1.' % i
            c = h.shape[1]
            theta = conv(a + 'theta', h, 0)
            phi = F.max_pool2d(conv(a + 'phi', h, 0), [2, 2])
            g = F.max_pool2d(conv(a + 'g', h, 0), [2, 2])
            n = h.shape[2] * h.shape[3]
            theta = theta.view(-1, c // 8, n)
            phi = phi.view(-1, c // 8, n // 4)
            g = g.view(-1, c // 2, n // 4)
            beta = F.softmax(torch.bmm(theta.transpose(1, 2), phi), -1)
            o = conv(a + 'o', torch.bmm(g, beta.transpose(1, 2)).view(-1, c // 2, h.shape[2], h.shape[3]), 0)
            h = sd[a + 'gamma'] * o + h
    h = F.batch_norm(h, sd['output_layer.0.stored_mean'], sd['output_layer.0.stored_var'], sd['output_layer.0.gain'],
                     sd['output_layer.0.bias'], False, 0.1, bn_eps)
    return torch.tanh(conv('output_layer.2', _relu(h), 1))
