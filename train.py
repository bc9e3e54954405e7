#!/usr/bin/env python
"""WarpedGANSpace training script on the MI355X-native path — same command line as the reference's
train.py:51-94 (flags, defaults, experiment directory, args.json), plus:
  --random-init-generator : keep the generator's constructor initialisation (no pre-trained files offline)
  --seed                  : seed of the device-side sampler
Multi-GPU: launch one process per GPU with torch.distributed.run; --batch-size is global.
"""
import argparse
import os

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC for RCCL on this driver

import torch
import torch.distributed as dist

from warpedganspace_amd.aux import create_exp_dir, exp_dir_name
from warpedganspace_amd.config import GAN_RESOLUTIONS, GAN_WEIGHTS, RECONSTRUCTOR_TYPES
from warpedganspace_amd.gan_load import build_gan
from warpedganspace_amd.reconstructor import Reconstructor
from warpedganspace_amd.support_sets import SupportSets
from warpedganspace_amd.trainer import Trainer


def parse(argv=None):
    p = argparse.ArgumentParser(description="WarpedGANSpace training script")
    p.add_argument('--gan-type', type=str, choices=GAN_WEIGHTS.keys(), help='set GAN generator model type')
    p.add_argument('--z-truncation', type=float, help="set latent code sampling truncation parameter")
    p.add_argument('--biggan-target-classes', nargs='+', type=int, help="list of classes for conditional BigGAN")
    p.add_argument('--stylegan2-resolution', type=int, default=1024, choices=(256, 1024), help="StyleGAN2 image resolution")
    p.add_argument('--shift-in-w-space', action='store_true', help="search latent paths in StyleGAN2's W-space")
    p.add_argument('-K', '--num-support-sets', type=int, help="set number of support sets (warping functions)")
    p.add_argument('-D', '--num-support-dipoles', type=int, help="set number of support dipoles per support set")
    p.add_argument('--learn-alphas', action='store_true', help='learn RBF alpha params')
    p.add_argument('--learn-gammas', action='store_true', help='learn RBF gamma params')
    p.add_argument('-g', '--gamma', type=float, help="set RBF gamma param (initial value when --learn-gammas)")
    p.add_argument('--support-set-lr', type=float, default=1e-4, help="set learning rate")
    p.add_argument('--reconstructor-type', type=str, choices=RECONSTRUCTOR_TYPES, default='ResNet')
    p.add_argument('--min-shift-magnitude', type=float, default=0.25)
    p.add_argument('--max-shift-magnitude', type=float, default=0.45)
    p.add_argument('--reconstructor-lr', type=float, default=1e-4)
    p.add_argument('--max-iter', type=int, default=100000)
    p.add_argument('--batch-size', type=int, default=32, help="GLOBAL batch size (split across ranks)")
    p.add_argument('--lambda-cls', type=float, default=1.00)
    p.add_argument('--lambda-reg', type=float, default=0.25)
    p.add_argument('--log-freq', default=10, type=int)
    p.add_argument('--ckp-freq', default=1000, type=int)
    p.add_argument('--tensorboard', action='store_true')
    p.add_argument('--cuda', dest='cuda', action='store_true')
    p.add_argument('--no-cuda', dest='cuda', action='store_false')
    p.set_defaults(cuda=True)
    ext = p.add_argument_group('extensions (not stored in args.json)')
    ext.add_argument('--random-init-generator', action='store_true')
    ext.add_argument('--seed', type=int, default=None)
    ext.add_argument('--precision', choices=('auto', 'fp32', 'fp32w', 'bf16x3', 'bf16x3w', 'f16', 'f16x2', 'mixed', 'mixed-strict'), default=None,
                     help="arithmetic of the frozen generator's convs (default: warpedganspace_amd.conv.DEFAULT_PRECISION = auto: the "
                          "cheapest mode measured inside the 1e-3 image-error gate for the architecture, else the fp32-class bf16x3; "
                          "fp32 = the reference's arithmetic everywhere)")
    ext.add_argument('--r-precision', choices=('auto', 'fp32', 'fp32w', 'bf16x3'), default='auto',
                     help="arithmetic of the trained reconstructor's convs: auto = exact fp32 with an fp32 generator, the fp32-class "
                          "split-bf16 x3 with a 16-bit one")
    ext.add_argument('--check-precision', type=int, default=0, metavar='N',
                     help="every N iterations regenerate the batch's images with the exact-fp32 kernels and report / check the "
                          "16-bit mode's image error against the 1e-3 gate (0 = off)")
    return p.parse_args(argv)


def main(argv=None):
    args = parse(argv)
    ext = {'random_init_generator': args.random_init_generator, 'seed': args.seed, 'precision': args.precision,
           'r_precision': args.r_precision, 'check_precision': args.check_precision}
    for k in ext:
        delattr(args, k)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local_rank)
        from warpedganspace_amd.hostpin import pin_rank        # this rank's launch thread stays on its GPU's NUMA node
        pin = pin_rank(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
        if rank == 0:
            print("#. Host placement of rank 0: {}".format(pin))
        dist.init_process_group('nccl', rank=rank, world_size=world)
    exp_dir = create_exp_dir(args) if rank == 0 else exp_dir_name(args)      # args.json: exactly the reference's keys
    if world > 1:
        dist.barrier()
    use_cuda = bool(args.cuda and torch.cuda.is_available())
    if not use_cuda:
        raise SystemExit("this build drives hand-written HIP kernels and needs an MI355X (--cuda); "
                         "the reference's CPU path exists here only as the test oracle")
    res = args.stylegan2_resolution if args.gan_type == 'StyleGAN2' else GAN_RESOLUTIONS[args.gan_type]
    weights = GAN_WEIGHTS[args.gan_type]['weights'][res]
    if rank == 0:
        print("#. Build GAN generator model G...")
        print("  \\__GAN type: {}".format(args.gan_type))
        print("  \\__Pre-trained weights: {}".format('<random init>' if ext['random_init_generator'] else weights))
    G = build_gan(args.gan_type, args.biggan_target_classes, args.stylegan2_resolution, args.shift_in_w_space, weights,
                  random_init=ext['random_init_generator'])
    S = SupportSets(num_support_sets=args.num_support_sets, num_support_dipoles=args.num_support_dipoles,
                    support_vectors_dim=G.dim_z, learn_alphas=args.learn_alphas, learn_gammas=args.learn_gammas,
                    gamma=1.0 / G.dim_z if args.gamma is None else args.gamma)
    R = Reconstructor(reconstructor_type=args.reconstructor_type, dim=S.num_support_sets,
                      channels=1 if args.gan_type == 'SNGAN_MNIST' else 3)
    if rank == 0:
        print("#. Support Sets: K={} N={} d={} trainable={:,}".format(
            args.num_support_sets, args.num_support_dipoles, G.dim_z, sum(p.numel() for p in S.parameters() if p.requires_grad)))
        print("#. Reconstructor trainable parameters: {:,}".format(sum(p.numel() for p in R.parameters() if p.requires_grad)))
        print("#. Experiment: {}".format(exp_dir))
    from warpedganspace_amd import conv as C
    args.seed = ext['seed']
    args.precision = ext['precision'] or C.DEFAULT_PRECISION
    args.r_precision = ext['r_precision']
    args.check_precision = ext['check_precision']
    trn = Trainer(params=args, exp_dir=exp_dir, use_cuda=use_cuda, multi_gpu=world > 1)
    trn.train(generator=G, support_sets=S, reconstructor=R)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
