"""CPU: a second, independently written float64 statement of torchvision's ResNet-18 feature extractor (numpy only: im2col by
stride tricks + matmul, explicit train-mode BatchNorm, explicit max-pool) against the oracle's (oracle/wgs_oracle.py::
resnet18_features, torch.nn.functional ops).

torchvision is an un-vendored dependency of the reference (lib/reconstructor.py:3,54-63) and absent from this image, so no
vector produced by torchvision itself can pin the oracle ("parity unpinned" for this sub-net).  This test is the
weaker substitute VERDICT r1 asked for: two statements written from the public definition —
    conv 7x7 / stride 2 / pad 3 (no bias) - BN - ReLU - max-pool 3x3 / stride 2 / pad 1
    - 4 stages x 2 BasicBlocks [ conv3x3(stride s) - BN - ReLU - conv3x3 - BN, + identity (or conv1x1(stride s) - BN when the
      shape changes), ReLU ], widths 64/128/256/512, s = 2 at the first block of stages 2-4 - global average pool —
with no shared code must agree to float64 round-off, including the order of the two convs' strides (torchvision >= 0.3
puts the stride on the FIRST 3x3 conv of a BasicBlock) and the shortcut's kernel size."""
import numpy as np
import torch

from oracle import wgs_oracle as O
from warpedganspace_amd.reconstructor import Reconstructor


def conv_np(x, w, stride, pad):
    """x [B,C,H,W], w [O,C,k,k] -> [B,O,Ho,Wo]; cross-correlation like nn.Conv2d, zero padding."""
    B, C, H, W = x.shape
    Oc, _, k, _ = w.shape
    xp = np.zeros((B, C, H + 2 * pad, W + 2 * pad))
    xp[:, :, pad:pad + H, pad:pad + W] = x
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    s = xp.strides
    win = np.lib.stride_tricks.as_strided(xp, (B, C, Ho, Wo, k, k), (s[0], s[1], s[2] * stride, s[3] * stride, s[2], s[3]))
    return np.einsum('bchwij,ocij->bohw', win, w, optimize=True)


def bn_train_np(x, gamma, beta, eps=1e-5):
    m = x.mean(axis=(0, 2, 3), keepdims=True)
    v = ((x - m) ** 2).mean(axis=(0, 2, 3), keepdims=True)          # biased variance normalises the batch
    return (x - m) / np.sqrt(v + eps) * gamma[None, :, None, None] + beta[None, :, None, None]


def maxpool_np(x, k=3, stride=2, pad=1):
    B, C, H, W = x.shape
    xp = np.full((B, C, H + 2 * pad, W + 2 * pad), -np.inf)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = np.full((B, C, Ho, Wo), -np.inf)
    for i in range(k):
        for j in range(k):
            out = np.maximum(out, xp[:, :, i:i + stride * Ho:stride, j:j + stride * Wo:stride])
    return out


def resnet18_np(sd, x):
    g = lambda k: sd['features_extractor.' + k].double().numpy()
    relu = lambda t: np.maximum(t, 0.0)
    h = relu(bn_train_np(conv_np(x, g('conv1.weight'), 2, 3), g('bn1.weight'), g('bn1.bias')))
    h = maxpool_np(h)
    width_in = 64
    for stage, width in enumerate((64, 128, 256, 512), start=1):
        for blk in range(2):
            s = 2 if (stage > 1 and blk == 0) else 1
            p = 'layer%d.%d.' % (stage, blk)
            y = relu(bn_train_np(conv_np(h, g(p + 'conv1.weight'), s, 1), g(p + 'bn1.weight'), g(p + 'bn1.bias')))
            y = bn_train_np(conv_np(y, g(p + 'conv2.weight'), 1, 1), g(p + 'bn2.weight'), g(p + 'bn2.bias'))
            if s != 1 or width_in != width:
                sc = bn_train_np(conv_np(h, g(p + 'downsample.0.weight'), s, 0), g(p + 'downsample.1.weight'), g(p + 'downsample.1.bias'))
            else:
                sc = h
            h = relu(y + sc)
            width_in = width
    return h.mean(axis=(2, 3))


def test_oracle_resnet18_equals_independent_numpy_statement():
    torch.manual_seed(3)
    R = Reconstructor('ResNet', 8)
    sd = {k: v.detach().clone() for k, v in R.state_dict().items()}
    for k in sd:                                   # non-trivial affine BN parameters
        if k.endswith('bn1.weight') or k.endswith('bn2.weight') or k.endswith('downsample.1.weight'):
            sd[k] = 1.0 + 0.3 * torch.randn_like(sd[k])
        if k.endswith('bn1.bias') or k.endswith('bn2.bias') or k.endswith('downsample.1.bias'):
            sd[k] = 0.2 * torch.randn_like(sd[k])
    x = torch.randn(3, 6, 64, 64, dtype=torch.float64)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    ref = O.resnet18_features(sd64, x, training=True).numpy()
    mine = resnet18_np(sd, x.numpy())
    assert mine.shape == (3, 512) == ref.shape
    err = np.abs(mine - ref).max() / np.abs(ref).max()
    assert err < 1e-10, err
    # manifest facts of the public definition the state_dict must show
    assert sd['features_extractor.layer2.0.downsample.0.weight'].shape == (128, 64, 1, 1)
    assert 'features_extractor.layer1.0.downsample.0.weight' not in sd
    assert sd['features_extractor.conv1.weight'].shape == (64, 6, 7, 7)
