#!/usr/bin/env python
"""WarpedGANSpace latent-space traversal on the MI355X-native path — same command line, inputs
(`<exp>/args.json`, `<exp>/models/support_sets*.pt`, `experiments/latent_codes/<gan>/<pool>/*/latent_code.pt`)
and outputs (`<exp>/results/<pool>/<2T>_<eps>_<len>/<hash>/{paths_images/path_%03d/%06d.jpg,
original_image.jpg, paths_latent_codes.pt}`) as the reference's traverse_latent_space.py:153-490.

All K paths x 2T steps of a latent code are integrated by ONE kernel launch (wgs_rbf_traverse, support
set LDS-resident) instead of K*2T sequential module calls; frames are rendered in generator batches.
Reference quirk kept: frame t is rendered at code_t + shift_t (the stored post-step code plus its last
shift, traverse_latent_space.py:457-462).
"""
import argparse
import json
import os
import os.path as osp

import numpy as np
import torch
from PIL import Image

from warpedganspace_amd.config import GAN_RESOLUTIONS, GAN_WEIGHTS
from warpedganspace_amd import conv as C
from warpedganspace_amd.gan_load import build_gan, set_generator_precision
from warpedganspace_amd.support_sets import SupportSets


def tensor2image(t, img_size=None):
    """Per-image min-max normalisation to uint8 (tensor2image(adaptive=True), traverse_latent_space.py:26-34)."""
    t = t.squeeze(0)
    t = (t - t.min()) / (t.max() - t.min())
    a = (255 * t.detach().cpu()).to(torch.uint8).numpy()
    img = Image.fromarray(a[0], mode='L') if a.shape[0] == 1 else Image.fromarray(np.transpose(a, (1, 2, 0)))
    return img.resize((img_size, img_size)) if img_size else img


def main(argv=None):
    p = argparse.ArgumentParser(description="WarpedGANSpace latent space traversal script")
    p.add_argument('-v', '--verbose', action='store_true')
    p.add_argument('--exp', type=str, required=True)
    p.add_argument('--pool', type=str, required=True)
    p.add_argument('--shift-steps', type=int, default=16)
    p.add_argument('--eps', type=float, default=0.2)
    p.add_argument('--shift-leap', type=int, default=1)
    p.add_argument('--batch-size', type=int)
    p.add_argument('--img-size', type=int)
    p.add_argument('--img-quality', type=int, default=75)
    p.add_argument('--gif', action='store_true')
    p.add_argument('--gif-size', type=int, default=256)
    p.add_argument('--gif-fps', type=int, default=30)
    p.add_argument('--cuda', dest='cuda', action='store_true')
    p.add_argument('--no-cuda', dest='cuda', action='store_false')
    p.add_argument('--random-init-generator', action='store_true', help="extension: no pre-trained generator file")
    p.add_argument('--pool-root', type=str, default=osp.join('experiments', 'latent_codes'))
    p.add_argument('--precision', choices=('auto', 'fp32', 'fp32w', 'bf16x3', 'bf16x3w', 'f16', 'f16x2', 'mixed', 'mixed-strict'), default=None,
                   help="extension: arithmetic of the generator's convs (default: the fp32-class bf16x3; fp32 = the reference's)")
    p.set_defaults(cuda=True)
    args = p.parse_args(argv)

    if not osp.isdir(args.exp):
        raise NotADirectoryError("Invalid given directory: {}".format(args.exp))
    args_json_file = osp.join(args.exp, 'args.json')
    if not osp.isfile(args_json_file):
        raise FileNotFoundError("File not found: {}".format(args_json_file))
    cfg = json.load(open(args_json_file))
    gan_type = cfg["gan_type"]
    models_dir = osp.join(args.exp, 'models')
    if not osp.isdir(models_dir):
        raise NotADirectoryError("Invalid models directory: {}".format(models_dir))
    ss_file = osp.join(models_dir, 'support_sets.pt')
    if not osp.isfile(ss_file):      # fall back to the latest support_sets-<iter>.pt (:201-208), by iteration NUMBER (the
        # reference's plain string sort ranks support_sets-9000.pt above support_sets-10000.pt)
        def _iter_of(f):
            digits = ''.join(ch for ch in f[len('support_sets-'):] if ch.isdigit())
            return int(digits) if digits else -1
        cands = sorted((f for f in os.listdir(models_dir) if f.startswith('support_sets-')), key=_iter_of)
        if not cands:
            raise FileNotFoundError("No support_sets.pt or support_sets-<iter>.pt under {}".format(models_dir))
        ss_file = osp.join(models_dir, cands[-1])
    pool = osp.join(args.pool_root, gan_type + ''.join('-{}'.format(c) for c in (cfg.get("biggan_target_classes") or []))
                    if gan_type == 'BigGAN' else gan_type, args.pool)
    if not osp.isdir(pool):
        raise NotADirectoryError("Invalid pool directory: {} -- Please run sample_gan.py to create it.".format(pool))
    if not (args.cuda and torch.cuda.is_available()):
        raise SystemExit("the traversal runs on the HIP kernels and needs an MI355X")
    dev = torch.device('cuda')

    res = cfg["stylegan2_resolution"] if gan_type == 'StyleGAN2' else GAN_RESOLUTIONS[gan_type]
    G = build_gan(gan_type, cfg.get("biggan_target_classes"), cfg["stylegan2_resolution"], cfg["shift_in_w_space"],
                  GAN_WEIGHTS[gan_type]['weights'][res], random_init=args.random_init_generator).to(dev).eval()
    set_generator_precision(G, args.precision or C.IMAGE_DEFAULT_PRECISION)
    S = SupportSets(num_support_sets=cfg["num_support_sets"], num_support_dipoles=cfg["num_support_dipoles"],
                    support_vectors_dim=G.dim_z, learn_alphas=cfg["learn_alphas"], learn_gammas=cfg["learn_gammas"],
                    gamma=1.0 / G.dim_z if cfg["gamma"] is None else cfg["gamma"])
    S.load_state_dict(torch.load(ss_file, map_location='cpu'))
    S.to(dev).eval()
    K, T, leap = S.num_support_sets, args.shift_steps, args.shift_leap
    out_dir = osp.join(args.exp, 'results', args.pool, '{}_{}_{}'.format(2 * T, args.eps, round(2 * T * args.eps, 3)))
    os.makedirs(out_dir, exist_ok=True)
    if args.batch_size is None:
        args.batch_size = 2 * T + 1

    code_dirs = sorted(d for d in os.listdir(pool) if osp.isdir(osp.join(pool, d)))
    zs = torch.cat([torch.load(osp.join(pool, d, 'latent_code.pt'), map_location='cpu') for d in code_dirs]).to(dev)
    w_space = bool(cfg["shift_in_w_space"])
    with torch.no_grad():
        start = G.get_w(zs) if w_space else zs
        path, shift = S.traverse(start.contiguous(), args.eps, T)          # [n, K, 2T+1, d] each, one launch
    keep = [T - j * leap for j in range(T // leap, 0, -1)] + [T] + [T + j * leap for j in range(1, T // leap + 1)]
    path, shift = path[:, :, keep], shift[:, :, keep]
    L = len(keep)
    for i, h in enumerate(code_dirs):
        code_dir = osp.join(out_dir, h)
        img_root = osp.join(code_dir, 'paths_images')
        os.makedirs(img_root, exist_ok=True)
        for k in range(K):
            codes, shifts = path[i, k], shift[i, k]
            frames = []
            with torch.no_grad():
                for a in range(0, L, args.batch_size):
                    cz, sh = codes[a:a + args.batch_size], shifts[a:a + args.batch_size]
                    frames.append(G(cz, sh, latent_is_w=True) if w_space else G(cz, sh))
            frames = torch.cat(frames)
            pdir = osp.join(img_root, 'path_{:03d}'.format(k))
            os.makedirs(pdir, exist_ok=True)
            pil = [tensor2image(frames[t], args.img_size) for t in range(L)]
            for t, im in enumerate(pil):
                im.save(osp.join(pdir, '{:06d}.jpg'.format(t)), "JPEG", quality=args.img_quality, optimize=True, progressive=True)
                if t == L // 2 and k == 0:
                    im.save(osp.join(code_dir, 'original_image.jpg'), "JPEG", quality=95, optimize=True, progressive=True)
            if args.gif:
                g = [im.resize((args.gif_size, args.gif_size)) for im in pil]
                g[0].save(osp.join(pdir, 'path.gif'), save_all=True, append_images=g[1:] + g[-2:0:-1],
                          duration=int(1000 / args.gif_fps), loop=0)
        torch.save(path[i].cpu(), osp.join(code_dir, 'paths_latent_codes.pt'))    # [K, L, d]
        if args.verbose:
            print("  \\__latent code {} [{}/{}] done".format(h, i + 1, len(code_dirs)))


if __name__ == '__main__':
    main()
