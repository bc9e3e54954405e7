"""Constants of the hot path (subset of the reference's lib/config.py:12-64; URLs / class-name tables
are not part of this path)."""
RECONSTRUCTOR_TYPES = ('ResNet', 'LeNet')

GAN_RESOLUTIONS = {'SNGAN_MNIST': 32, 'SNGAN_AnimeFaces': 64, 'BigGAN': 128, 'ProgGAN': 1024, 'StyleGAN2': 1024}

GAN_WEIGHTS = {
    'SNGAN_MNIST': {'weights': {32: 'models/pretrained/generators/SNGAN_MNIST/generator.pt'}},
    'SNGAN_AnimeFaces': {'weights': {64: 'models/pretrained/generators/SNGAN_AnimeFaces/generator.pt'}},
    'BigGAN': {'weights': {128: 'models/pretrained/generators/BigGAN/G_ema.pth'}},
    'ProgGAN': {'weights': {1024: 'models/pretrained/generators/ProgGAN/100_celeb_hq_network-snapshot-010403.pth'}},
    'StyleGAN2': {'weights': {256: 'models/pretrained/generators/StyleGAN2/stylegan2-ffhq-256-550000.pt',
                              1024: 'models/pretrained/generators/StyleGAN2/stylegan2-ffhq-config-f.pt'}},
}
