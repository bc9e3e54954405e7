"""GPU: every arithmetic mode of the generator convs (fp32 / bf16x3 / f16 / f16x2) end to end against the float64 oracle,
at 256x256 (BASELINE configs[2]) and 1024x1024 (configs[4], which names the fp16 MFMA path):

  * image error (max-norm relative, the north_star's gate: 1e-3);
  * input-gradient error with the oracle forced through the SAME leaky-relu gates the HIP forward took (the only
    well-posed gradient comparison: every remaining difference is arithmetic, not gate flips), W space (the synthesis
    network — what the conv arithmetic touches) and Z space (adds the ill-conditioned random mapping net).

A mode that misses the 1e-3 line on the image or the W-space gradient is REPORTED and must not be the default mode
(conv.DEFAULT_PRECISION); the numbers are also written to gpurun_out/precision_schemes.json for DESIGN.md."""
import json
import os

import pytest
import torch

from oracle import wgs_oracle as O
from tests import golden_inputs as GI
from tests.util import rel_err
from warpedganspace_amd import conv as C
from warpedganspace_amd.gan_load import StyleGAN2Wrapper

pytestmark = pytest.mark.gpu
GATE = 1e-3
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(key, val):
    path = os.path.join(REPO, 'gpurun_out', 'precision_schemes.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    d = json.load(open(path)) if os.path.exists(path) else {}
    d[key] = val
    json.dump(d, open(path, 'w'), indent=1, sort_keys=True)


@pytest.mark.parametrize('size,B', [(256, 2), (1024, 1)])
def test_image_and_shared_gate_gradient_per_scheme(dev, size, B):
    from tests.test_stylegan2_gpu import build
    G, sd = build(size, 4000 + size, dev)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    z = GI.rt(11 + size, B, 512)
    old = C.PRECISION
    rows = {}
    try:
        for name in ('fp32', 'bf16x3', 'f16', 'f16x2'):
            C.set_precision(name)
            res = {}
            for w_space in (True, False):
                G.debug_keep = {}
                sh = (GI.rt(12 + size, B, 512) * 0.1).to(dev).requires_grad_(True)
                img = StyleGAN2Wrapper(G, w_space)(z.to(dev), sh)
                probe = GI.rt(13 + size, *img.shape)
                (img * probe.to(dev)).sum().backward()
                gates = ([] if w_space else [g.cpu() for g in G.debug_keep['mapping']]) + [g.cpu() for g in G.debug_keep['synthesis']]
                G.debug_keep = None
                sho = (GI.rt(12 + size, B, 512) * 0.1).double().requires_grad_(True)
                if w_space:
                    w = O.sg2_mapping(sd64, z.double()).detach()
                    O.GATE_OVERRIDE = iter(gates)
                    img_o = O.sg2_synthesis(sd64, w + sho, size)
                else:
                    O.GATE_OVERRIDE = iter(gates)
                    img_o = O.sg2_generate(sd64, z.double(), size, sho)
                O.GATE_OVERRIDE = None
                (img_o * probe.double()).sum().backward()
                tag = 'W' if w_space else 'Z'
                res['img_' + tag] = rel_err(img, img_o.detach())
                res['grad_' + tag] = rel_err(sh.grad, sho.grad)
                del img_o, sho
            rows[name] = res
            print('StyleGAN2-%d %-7s image err W %.2e Z %.2e | shared-gate gradient err W %.2e Z %.2e' % (
                size, name, res['img_W'], res['img_Z'], res['grad_W'], res['grad_Z']))
    finally:
        C.PRECISION = old
    _record('stylegan2_%d' % size, rows)
    for name, res in rows.items():
        ok = res['img_W'] < GATE and res['img_Z'] < GATE and res['grad_W'] < GATE
        if name in ('fp32', 'bf16x3'):
            assert res['img_W'] < 1e-4 and res['grad_W'] < 2e-4, (name, res)
        if not ok:
            assert name != C.DEFAULT_PRECISION, "default arithmetic %s misses the 1e-3 gate: %r" % (name, res)
    # whatever the mode: no NaN / inf, and never worse than a few fp16 ulps
    for name, res in rows.items():
        assert all(v == v and v < 1e-2 for v in res.values()), (name, res)


@pytest.mark.parametrize('name', ['f16', 'f16x2'])
def test_step_loss_and_argmax_fp16_schemes(dev, name):
    """One training step (StyleGAN2-32, K=16, ResNet-18 R) in the fp16 modes against the oracle's replay: loss within 1e-3,
    path-index argmax bit-exact, S / R gradients within a few 1e-3 (fp16 operand rounding: 2^-11)."""
    from tests.test_train_step_gpu import make
    old = C.set_precision(name)
    try:
        eng, ref, c = make(dev, 32, 16, 4, 4, False)
        g = torch.Generator().manual_seed(7)
        z = torch.randn(4, 512, generator=g)
        idx = torch.randint(0, 16, (4,), generator=g)
        mag = (torch.rand(4, generator=g) * 0.2 + 0.25)
        o = ref.step(z, idx, mag)
        st = eng.step(z.to(dev), idx.to(dev), mag.to(dev)).tolist()
        gb = eng.bucket.gview
        e_s = rel_err(gb[id(eng.S.SUPPORT_SETS)], ref.s['SUPPORT_SETS'].grad)
        e_r = max(rel_err(prm.grad, ref.r[n].grad) for n, prm in eng.R.named_parameters() if not n.startswith('features_extractor.fc'))
        print('%s step: loss %.6f (oracle %.6f), dS err %.2e, worst dR err %.2e' % (name, st[2], o['loss'], e_s, e_r))
        assert abs(st[2] - o['loss']) < 1e-3 * max(1.0, abs(o['loss']))
        assert torch.equal(eng.argmax.cpu(), o['argmax'])
        assert e_s < 2e-2 and e_r < 2e-2
    finally:
        C.PRECISION = old
