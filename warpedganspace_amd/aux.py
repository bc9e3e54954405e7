"""Helpers of the training driver — mirror of the hot-path parts of the reference's lib/aux.py:
`sample_z` (:39-53), `TrainingStatTracker` (:13-36), `create_exp_dir` (:56-104), progress helpers."""
import json
import os
import os.path as osp
import sys

import numpy as np
import torch


class TrainingStatTracker(object):
    def __init__(self):
        self.stat_tracker = {'accuracy': [], 'classification_loss': [], 'regression_loss': [], 'total_loss': []}

    def update(self, accuracy, classification_loss, regression_loss, total_loss):
        self.stat_tracker['accuracy'].append(float(accuracy))
        self.stat_tracker['classification_loss'].append(float(classification_loss))
        self.stat_tracker['regression_loss'].append(float(regression_loss))
        self.stat_tracker['total_loss'].append(float(total_loss))

    def get_means(self):
        return {key: np.mean(value) for key, value in self.stat_tracker.items()}

    def flush(self):
        for key in self.stat_tracker.keys():
            self.stat_tracker[key] = []


def sample_z(batch_size, dim_z, truncation=None, device=None, generator=None):
    """Latent codes ~ N(0, I), optionally truncated to [-truncation, truncation] (lib/aux.py:39-53).
    Extension: `device` samples directly in HBM (no host round trip); the truncated case draws
    with rejection-free inverse-CDF sampling on the device instead of scipy on the host."""
    if truncation is None or truncation == 1.0:
        return torch.randn(batch_size, dim_z, device=device, generator=generator)
    # truncated standard normal via inverse CDF: Phi^-1( Phi(-t) + u * (Phi(t) - Phi(-t)) )
    u = torch.rand(batch_size, dim_z, device=device, generator=generator, dtype=torch.float64)
    lo = 0.5 * (1 + torch.erf(torch.tensor(-truncation / 2 ** 0.5, dtype=torch.float64)))
    hi = 0.5 * (1 + torch.erf(torch.tensor(truncation / 2 ** 0.5, dtype=torch.float64)))
    p = (lo + u * (hi - lo)).clamp(1e-12, 1 - 1e-12)
    return (2 ** 0.5 * torch.erfinv(2 * p - 1)).to(torch.float32)


def exp_dir_name(args):
    """Experiment directory name, lib/aux.py:60-90:
    <gan_type>(-<res>-{Z,W})(-<classes>)-<R type>-K<K>-D<N>(-LearnAlphas)(-LearnGammas)-eps<min>_<max>"""
    exp_dir = "{}".format(args.gan_type)
    if args.gan_type == 'StyleGAN2':
        exp_dir += '-{}'.format(args.stylegan2_resolution)
        exp_dir += '-W' if args.shift_in_w_space else '-Z'
    if args.gan_type == 'BigGAN':
        exp_dir += '-' + ''.join('{}'.format(c) for c in args.biggan_target_classes)
    exp_dir += "-{}".format(args.reconstructor_type)
    exp_dir += "-K{}-D{}".format(args.num_support_sets, args.num_support_dipoles)
    if args.learn_alphas:
        exp_dir += '-LearnAlphas'
    if args.learn_gammas:
        exp_dir += '-LearnGammas'
    exp_dir += "-eps{}_{}".format(args.min_shift_magnitude, args.max_shift_magnitude)
    return exp_dir


def create_exp_dir(args, root="experiments"):
    """Create experiments/wip/<exp_dir>/ with args.json and command.sh (lib/aux.py:92-104)."""
    exp_dir = exp_dir_name(args)
    wip_dir = osp.join(root, "wip", exp_dir)
    os.makedirs(wip_dir, exist_ok=True)
    with open(osp.join(wip_dir, 'args.json'), 'w') as f:
        json.dump(args.__dict__, f)
    with open(osp.join(wip_dir, 'command.sh'), 'w') as f:
        f.write('#!/usr/bin/bash\n')
        f.write(' '.join(sys.argv) + '\n')
    return exp_dir


def update_progress(msg, total, progress, width=20):
    """One-line console progress bar (cosmetic; the training log of lib/trainer.py uses the same 20-cell bar)."""
    frac = min(max(float(progress) / float(total), 0.0), 1.0)
    cells = int(round(width * frac))
    done = frac >= 1.0
    sys.stdout.write("\r{}{}{} {:.0f}% {}".format(msg, "\u2588" * cells, "\u2591" * (width - cells), round(100 * frac), "\r\n" if done else ""))
    sys.stdout.flush()


def update_stdout(num_lines):
    """Move the cursor up over the block the log just printed, so that the next log overwrites it (interactive terminals only)."""
    sys.stdout.write('\x1b[1A\x1b[1A\n' * num_lines)       # up two, down one (the newline): one line up per repetition
    sys.stdout.flush()


_UNITS = (("days", 86400), ("hours", 3600), ("minutes", 60))


def sec2dhms(t):
    """Seconds -> 'DD days, HH hours, MM minutes, and SS seconds' (the reference log's format)."""
    parts, t = [], int(t)
    for name, size in _UNITS:
        q, t = divmod(t, size)
        parts.append("%02d %s" % (q, name))
    return ", ".join(parts) + ", and %02d seconds" % t
