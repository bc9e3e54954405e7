"""Generator wrappers with the reference's uniform API `G(z, shift=None)`, `G.dim_z`, `G.get_w(z)`
(models/gan_load.py:21-28,65-81,109-120,137-179) on top of the HIP generators.

Pre-trained weights are not available offline; `build_*` accept `pretrained_gan_weights=None`
(random init exactly as the reference constructors do) or a path in the reference's file formats.
"""
import torch
from torch import nn

from .stylegan2 import Generator as StyleGAN2Generator


class StyleGAN2Wrapper(nn.Module):
    """models/gan_load.py:137-179."""

    def __init__(self, G, shift_in_w_space):
        super().__init__()
        self.G = G
        self.shift_in_w_space = shift_in_w_space
        self.dim_z = 512
        self.dim_w = self.G.style_dim if self.shift_in_w_space else self.dim_z

    def get_w(self, z):
        """Z-space codes [B,512] -> W-space codes [B,512] (mapping network)."""
        return self.G.get_latent(z)

    def forward(self, z, shift=None, latent_is_w=False, precision=None, policy=None):
        """z: latent codes (Z space, or W space when `latent_is_w`); shift: shift vectors in the space
        selected by `shift_in_w_space`.  Returns images [B, 3, res, res] (NCHW, un-clamped).
        precision (extension): arithmetic of this call's convs (conv.PRECISION_NAMES); default: the generator's `precision`.
        policy (extension): conv.MixedPolicy of this call under a 'mixed' mode (a step engine passes the table it calibrated)."""
        kw = dict(precision=precision, policy=policy)
        if self.shift_in_w_space:
            if latent_is_w:
                return self.G([z if shift is None else z + shift], input_is_latent=True, **kw)[0]
            w = self.G.get_latent(z)
            return self.G([w if shift is None else w + shift], input_is_latent=True, **kw)[0]
        return self.G([z if shift is None else z + shift], input_is_latent=False, **kw)[0]

    def resolve_precision(self, requested=None):
        return self.G.resolve_precision(requested)

    # -- the un-shifted pass G(z) in two stages (extension; trainer.TrainStep) ------------------------------------------------------
    def begin(self, z, precision=None, pause_res=32, policy=None):
        """Enqueue the mapping network and the synthesis layers up to `pause_res` for G(z) (no shift, nothing saved): a handle."""
        w, _ = self.G._mapping_fwd(z, save=False)
        return self.G.synthesis_begin(w, self.G.resolve_precision(precision), pause_res, policy)

    def advance(self, handle):
        """Enqueue the layers up to the next pause resolution (`pause_res` a tuple): None while paused again, else the image."""
        return self.G.synthesis_advance(handle)

    def finish(self, handle):
        """Enqueue the remaining layers: the image [B, 3, res, res]."""
        return self.G.synthesis_finish(handle)


def set_generator_precision(G, precision):
    """Arithmetic of the convs of wrapper / generator `G` for calls that do not pass `precision=` (conv.PRECISION_NAMES):
    an attribute of that generator instance, nothing process-wide.  Returns G."""
    from . import conv as C
    C.precision_code(precision)                  # validates the name
    inner = G.G if hasattr(G, 'G') else G
    inner.precision = precision
    return G


def build_stylegan2(pretrained_gan_weights=None, resolution=1024, shift_in_w_space=False):
    """models/gan_load.py:182-188 (checkpoint key 'g_ema', strict=False)."""
    G = StyleGAN2Generator(resolution, 512, 8)
    if pretrained_gan_weights is not None:
        G.load_state_dict(torch.load(pretrained_gan_weights, map_location='cpu')['g_ema'], strict=False)
    return StyleGAN2Wrapper(G, shift_in_w_space=shift_in_w_space)


def build_gan(gan_type, target_classes=None, stylegan2_resolution=1024, shift_in_w_space=False, weights=None,
              random_init=False):
    """Dispatcher used by train.py / traverse_latent_space.py (traverse_latent_space.py:43-69).
    `weights` = path of the pre-trained generator file (lib/config.py GAN_WEIGHTS); with `random_init` the
    generator keeps its constructor initialisation (no checkpoints are available offline)."""
    import os
    if weights is not None and not random_init and not os.path.isfile(weights):
        raise FileNotFoundError("pre-trained generator weights not found: {} (use --random-init-generator for a "
                                "synthetic run)".format(weights))
    w = None if random_init else weights
    if gan_type == 'StyleGAN2':
        return build_stylegan2(w, resolution=stylegan2_resolution, shift_in_w_space=shift_in_w_space)
    if gan_type == 'ProgGAN':
        from .proggan import build_proggan
        return build_proggan(w)
    if gan_type in ('SNGAN_MNIST', 'SNGAN_AnimeFaces'):
        from .sngan import build_sngan
        return build_sngan(w, gan_type)
    if gan_type == 'BigGAN':
        from .biggan import build_biggan
        return build_biggan(w, target_classes)
    raise ValueError("unknown gan_type {!r}".format(gan_type))
