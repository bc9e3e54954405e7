#!/usr/bin/env python
"""Latent-code pool producer on the MI355X-native generators — the command line and the on-disk layout of the reference's
sample_gan.py (:52-65 flags, :70-91 pool directory + args.json, :156-179 per-code files), which
traverse_latent_space.py consumes:

    experiments/latent_codes/<gan_type>[-<class>...]/<pool or <gan_type>_<num_samples>>/
        args.json
        <sha1 of the code's float32 bytes>/latent_code.pt   float32 [1, dim_z]
        <sha1 ...>/image.jpg                                 per-image min-max normalised JPEG (quality 95)

Differences by design: the codes are rendered in batches (--batch-size, default 32) instead of one generator call per code
(same images: the generators are batch-independent in eval mode — tests/test_stylegan2_gpu.py::test_full_size_batch_consistency),
and --random-init-generator / --seed / --precision exist because no pre-trained weights are available offline; they are
not written to args.json.
"""
import argparse
import json
import os
import os.path as osp
from hashlib import sha1

import torch

from warpedganspace_amd.aux import sample_z, update_progress
from warpedganspace_amd.config import GAN_RESOLUTIONS, GAN_WEIGHTS
from warpedganspace_amd import conv as C
from warpedganspace_amd.gan_load import build_gan, set_generator_precision


def tensor2image(t):
    """[3 or 1, H, W] float -> PIL image with per-image min-max normalisation (tensor2image(adaptive=True), sample_gan.py:13-17)."""
    from PIL import Image
    t = t.detach().float().cpu()
    t = (t - t.min()) / (t.max() - t.min())
    a = (255 * t).to(torch.uint8).permute(1, 2, 0).numpy()
    return Image.fromarray(a[:, :, 0], mode='L') if a.shape[2] == 1 else Image.fromarray(a, mode='RGB')


def parse(argv=None):
    p = argparse.ArgumentParser(description="Sample a pre-trained GAN latent space and generate images")
    p.add_argument('-v', '--verbose', action='store_true', help="set verbose mode on")
    p.add_argument('-g', '--gan-type', type=str, required=True, choices=GAN_WEIGHTS.keys(), help='GAN generator model type')
    p.add_argument('--shift-in-w-space', action='store_true', help="search latent paths in StyleGAN2's W-space")
    p.add_argument('--z-truncation', type=float, help="set latent code sampling truncation parameter")
    p.add_argument('--biggan-target-classes', nargs='+', type=int, help="list of classes for conditional BigGAN")
    p.add_argument('--stylegan2-resolution', type=int, default=1024, choices=(256, 1024), help="StyleGAN2 image resolution")
    p.add_argument('--num-samples', type=int, default=4, help="number of latent codes to sample")
    p.add_argument('--pool', type=str, help="name of latent codes/images pool")
    p.add_argument('--cuda', dest='cuda', action='store_true')
    p.add_argument('--no-cuda', dest='cuda', action='store_false')
    p.set_defaults(cuda=True)
    ext = p.add_argument_group('extensions (not stored in args.json)')
    ext.add_argument('--random-init-generator', action='store_true')
    ext.add_argument('--seed', type=int, default=None, help="seed of the latent-code sampler")
    ext.add_argument('--batch-size', type=int, default=32, help="codes rendered per generator call")
    ext.add_argument('--precision', choices=('auto', 'fp32', 'fp32w', 'bf16x3', 'bf16x3w', 'f16', 'f16x2', 'mixed', 'mixed-strict'), default=None,
                     help="arithmetic of the generator's convs (default: the fp32-class bf16x3; fp32 = the reference's)")
    ext.add_argument('--root', type=str, default='experiments', help="root of the experiments tree")
    return p, p.parse_args(argv)


def main(argv=None):
    parser, args = parse(argv)
    ext = {k: getattr(args, k) for k in ('random_init_generator', 'seed', 'batch_size', 'precision', 'root')}
    for k in ext:
        delattr(args, k)
    out_dir = osp.join(ext['root'], 'latent_codes', args.gan_type)
    classes = ''
    if args.gan_type == 'BigGAN':
        if args.biggan_target_classes is None:
            parser.error("In case of BigGAN, a list of classes needs to be determined.")
        classes = ''.join('-{}'.format(c) for c in args.biggan_target_classes)
        out_dir += classes
    out_dir = osp.join(out_dir, args.pool if args.pool else '{}_{}'.format(args.gan_type + classes, args.num_samples))
    os.makedirs(out_dir, exist_ok=True)
    with open(osp.join(out_dir, 'args.json'), 'w') as f:
        json.dump(args.__dict__, f)                                            # exactly the reference's keys
    if not (args.cuda and torch.cuda.is_available()):
        raise SystemExit("sample_gan.py renders with the HIP generators and needs an MI355X (--cuda)")
    dev = torch.device('cuda')
    res = args.stylegan2_resolution if args.gan_type == 'StyleGAN2' else GAN_RESOLUTIONS[args.gan_type]
    weights = GAN_WEIGHTS[args.gan_type]['weights'][res]
    if args.verbose:
        print("#. Build GAN generator model G and load with pre-trained weights...")
        print("  \\__GAN type: {}".format(args.gan_type))
        if args.gan_type == 'BigGAN':
            print("      \\__Target classes: {}".format(args.biggan_target_classes))
        print("  \\__Pre-trained weights: {}".format('<random init>' if ext['random_init_generator'] else weights))
    G = build_gan(args.gan_type, args.biggan_target_classes, args.stylegan2_resolution, args.shift_in_w_space, weights,
                  random_init=ext['random_init_generator']).to(dev).eval()
    set_generator_precision(G, ext['precision'] or C.IMAGE_DEFAULT_PRECISION)
    if args.verbose:
        print("#. Sample {} {}-dimensional latent codes...".format(args.num_samples, G.dim_z))
        if args.z_truncation:
            print("  \\__Truncate standard Gaussian to range [{}, +{}]".format(-args.z_truncation, args.z_truncation))
    gen = None
    if ext['seed'] is not None:
        gen = torch.Generator(device=dev).manual_seed(ext['seed'])
    zs = sample_z(args.num_samples, G.dim_z, truncation=args.z_truncation, device=dev, generator=gen)
    if args.verbose:
        print("#. Generate images...")
        print("  \\__{}".format(out_dir))
    done = 0
    for lo in range(0, args.num_samples, ext['batch_size']):
        zb = zs[lo:lo + ext['batch_size']]
        with torch.no_grad():
            imgs = G(zb).cpu()
        for i in range(zb.shape[0]):
            z = zb[i:i + 1].cpu()
            code_hash = sha1(z.numpy()).hexdigest()                            # sample_gan.py:159
            d = osp.join(out_dir, code_hash)
            os.makedirs(d, exist_ok=True)
            torch.save(z, osp.join(d, 'latent_code.pt'))                       # float32 [1, dim_z], :171
            tensor2image(imgs[i]).save(osp.join(d, 'image.jpg'), "JPEG", quality=95, optimize=True, progressive=True)
            done += 1
            if args.verbose:
                update_progress("  \\__.Latent code hash: {} [{:03d}/{:03d}] ".format(code_hash, done, args.num_samples),
                                args.num_samples, done)
    if args.verbose:
        print()
    return out_dir


if __name__ == '__main__':
    main()
