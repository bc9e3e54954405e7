#!/usr/bin/env python
"""Split experiments/wip/<exp>/models/checkpoint.pt into support_sets-<iter>.pt / reconstructor-<iter>.pt
(same behaviour and file names as the reference's checkpoint2model.py:37-49)."""
import argparse
import os.path as osp

import torch


def main(argv=None):
    ap = argparse.ArgumentParser(description="Convert a checkpoint file into support sets / reconstructor weight files")
    ap.add_argument('--exp', type=str, required=True, help="experiment dir created by train.py")
    args = ap.parse_args(argv)
    if not osp.isdir(args.exp):
        raise NotADirectoryError("Invalid given directory: {}".format(args.exp))
    models_dir = osp.join(args.exp, 'models')
    if not osp.isdir(models_dir):
        raise NotADirectoryError("Invalid models directory: {}".format(models_dir))
    ckpt = osp.join(models_dir, 'checkpoint.pt')
    if not osp.isfile(ckpt):
        raise FileNotFoundError("Checkpoint file not found: {}".format(ckpt))
    d = torch.load(ckpt, map_location='cpu')
    it = d['iter']
    print("#. Checkpoint iteration: {}".format(it))
    torch.save(d['support_sets'], osp.join(models_dir, 'support_sets-{}.pt'.format(it)))
    torch.save(d['reconstructor'], osp.join(models_dir, 'reconstructor-{}.pt'.format(it)))


if __name__ == '__main__':
    main()
