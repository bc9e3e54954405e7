"""Training driver — mirror of the reference's lib/trainer.py (`Trainer(params, exp_dir, use_cuda,
multi_gpu).train(generator, support_sets, reconstructor)`) around an MI355X-native step.

`TrainStep` is the hot loop body (lib/trainer.py:190-261) re-scheduled for one process per GPU:
  * z, path indices and shift magnitudes are sampled in HBM (same distributions, incl. the reference's
    arange-weighted draw without replacement, :212-221) — no host round trips, no per-sample mask loop;
  * the un-shifted branch G(z) runs without saving anything (it has no trainable ancestor); only the
    shifted branch is differentiated, and only w.r.t. its input (the reference also builds G's weight
    gradients and throws them away);
  * R's and S's gradients land in ONE flat fp32 bucket [R | S] that is all-reduced (RCCL, sum) once
    per step when world_size > 1 and consumed by two fused Adam launches (grad_scale = 1/world);
  * loss, its gradient, argmax and accuracy come from one kernel; statistics stay on the device
    until a log boundary.
Data-parallel semantics follow SURVEY.md §5.8: `--batch-size` is GLOBAL (local = global/world), the loss
is a mean over the global batch, BatchNorm statistics stay per rank, rank 0 writes checkpoints.
"""
import json
import os
import os.path as osp
import shutil
import sys
import time

import numpy as np
import ctypes

import torch
import torch.distributed as dist

from . import _lib as L
from . import conv as C
from .aux import TrainingStatTracker, sample_z, sec2dhms, update_progress, update_stdout
from .support_sets import rbf_workspace


def sampler_seed(seed, rank, world, start_iter, device):
    """Seed of this rank's device-side sampler.  Every rank must draw DIFFERENT (z, idx, magnitude) samples (the global
    batch is the union of the ranks' local batches), an unseeded run must not repeat the previous run's sequence (the
    reference samples from torch's unseeded global RNG, lib/trainer.py:195-221), and a resumed run must not replay the
    sampler from its start.  base = --seed if given, else a fresh random value drawn on rank 0 and broadcast."""
    if seed is None:
        base = int.from_bytes(os.urandom(5), 'little')       # 40 random bits; torch's global RNG is left alone
        if world > 1 and dist.is_available() and dist.is_initialized():
            t = torch.tensor([base], dtype=torch.int64, device=device if dist.get_backend() == 'nccl' else 'cpu')
            dist.broadcast(t, 0)
            base = int(t.item())
    else:
        base = int(seed)
    return ((base * 1000003 + 7919 * int(start_iter)) * max(world, 1) + rank) % (1 << 63)     # torch.Generator seeds are < 2^64


class _EagerSide:
    """Runs queued closures at once on `side`, behind what the current stream has enqueued so far (list-like: .append((x, dy, fn)))."""

    def __init__(self, side):
        self.side = side

    def append(self, item):
        x, dy, fn = item
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            fn()
        x.record_stream(self.side); dy.record_stream(self.side)


_STREAMS = {}


def _engine_streams(device):
    """(side, prefetch, high-priority main) streams of `device`, created once per process."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _STREAMS:
        # (CU-masked side streams — hipExtStreamCreateWithCUMask, 3/4 or 1/2 of every XCD's CUs, so that the critical path's short
        # launches always find free CUs — measured slower both for all side work (auto 25.7 -> 26.7 / 30.0 ms, fp32w 53.7 -> 59.9 / 69.9)
        # and for the prefetched pass's middle stage alone, the one that runs beside the Reconstructor (auto 25.7 -> 26.5 / 38.9 ms))
        _STREAMS[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device), torch.cuda.Stream(device=device, priority=-1))
    return _STREAMS[key]


class FlatBucket:
    """Re-homes parameters into one flat fp32 buffer (memory order preserved) with a matching flat gradient
    and Adam moment buffers.  `gview[id(p)]` is the gradient region of p in p's MEMORY layout (conv weights:
    packed [Co, kh*kw, Ci])."""

    def __init__(self, groups, device):
        # groups: list of (lr, [params])
        self.segments, self.gview, self.groups = [], {}, []
        total = 0
        for lr, params in groups:
            start = total
            for p in params:
                n = (p.numel() + 3) & ~3
                self.segments.append((p, total, p.numel()))
                total += n
            self.groups.append((lr, start, total))
        self.flat = torch.zeros(total, device=device)
        self.grad = torch.zeros(total, device=device)
        self.exp_avg = torch.zeros(total, device=device)
        self.exp_avg_sq = torch.zeros(total, device=device)
        self.step_count = 0
        with torch.no_grad():
            for p, off, n in self.segments:
                seg, gseg = self.flat[off:off + n], self.grad[off:off + n]
                if p.dim() == 4:
                    Co, Ci, kh, kw = p.shape
                    seg.view(Co, kh, kw, Ci).copy_(p.data.permute(0, 2, 3, 1))
                    p.data = seg.view(Co, kh, kw, Ci).permute(0, 3, 1, 2)
                    self.gview[id(p)] = gseg.view(Co, kh * kw, Ci)
                    p.grad = gseg.view(Co, kh, kw, Ci).permute(0, 3, 1, 2)
                else:
                    seg.view(p.shape).copy_(p.data)
                    p.data = seg.view(p.shape)
                    self.gview[id(p)] = gseg.view(p.shape)
                    p.grad = gseg.view(p.shape)

    def zero_grad(self):
        self.grad.zero_()

    def adam_step(self, world=1, betas=(0.9, 0.999), eps=1e-8):
        self.step_count += 1
        for lr, a, b in self.groups:
            if b > a:
                L.check(L.lib().wgs_adam_step(L.rawptr(self.flat[a:]), L.rawptr(self.grad[a:]), L.rawptr(self.exp_avg[a:]),
                                              L.rawptr(self.exp_avg_sq[a:]), L.c_int64(b - a), L.c_float(lr),
                                              L.c_float(betas[0]), L.c_float(betas[1]), L.c_float(eps), self.step_count,
                                              L.c_float(1.0 / world), L.stream()), 'wgs_adam_step')


class TrainStep:
    """One optimisation step of lib/trainer.py:190-261 on this rank's share of the batch."""

    def __init__(self, generator, support_sets, reconstructor, params, local_batch, device, world=1, seed=None, rank=0,
                 start_iter=0, precision=None, r_precision='auto', two_streams=True, defer_wgrad=True, prefetch=True, priority_main=True,
                 calibrate_images=None):
        """precision: arithmetic of the frozen generator's convs (conv.PRECISION_NAMES; None = the generator's own `precision`
        attribute, whose default is the reference's fp32).  r_precision: 'fp32' | 'bf16x3' | 'auto' | reconstructor.RArith —
        arithmetic of the trained Reconstructor's convs ('auto': exact fp32 when the generator runs exact fp32, the fp32-class
        split-bf16 x3 when it runs in a 16-bit mode).  Both belong to THIS engine: engines with different modes coexist."""
        from .reconstructor import r_arith
        self.G, self.S, self.R, self.p = generator, support_sets, reconstructor, params
        self.B, self.dev, self.world, self.rank = local_batch, device, world, rank
        self.precision = generator.resolve_precision(precision)       # concrete code 0..4
        self.r_arith = r_arith(r_precision, self.precision)
        self.gen = torch.Generator(device=device)
        # HIP maps streams onto a handful of hardware queues (4 per process by default); every engine of a process making its own
        # three streams put the fourth engine's "concurrent" streams on ONE queue — bench.py's second engine ran 37.5 instead of
        # 27.5 ms per step.  The streams are per device, shared by all engines.
        self.side_stream, self.pre_stream, main_stream = _engine_streams(device)
        # The step's critical path (shifted forward -> R -> loss -> R backward -> G backward -> Adam) runs on a HIGH-priority stream of
        # its own; the side work that only has to be finished by the end of the step (the un-shifted pass of this / the next batch,
        # R's weight gradients) stays on default-priority streams: when a CU frees up, the critical path's next workgroups — the
        # Reconstructor's ~300 short launches in particular — are dispatched ahead of the side streams' long-running generator tiles
        # instead of queueing behind them.
        self.main_stream = main_stream if (priority_main and two_streams) else None
        self.two_streams = two_streams       # un-shifted generator pass on the side stream
        self.defer_wgrad = defer_wgrad       # R's weight gradients on the side stream, next to the generator's backward
        self.prefetch = prefetch             # G(z) of the NEXT step's batch, next to this step's Reconstructor / backward phases
        self.eager_wgrad = getattr(TrainStep, 'eager_wgrad_default', False)     # R's weight gradients next to R's own backward (measured: 26.47 -> 26.71 ms, off)
        self.split_pause_res = getattr(TrainStep, 'split_pause_res_default', 32)
        self.split_prefetch = getattr(TrainStep, 'split_prefetch_default', True)           # ... its low-resolution layers already next to this step's shifted forward (StyleGAN2)
        # ... and its last, chip-filling layers (above tail_pause_res) held back until the generator's BACKWARD reaches its latency-bound
        # layers (<= tail_hook_res): they fill the chip under the backward's tail, the RBF backward, Adam and the head of the next step
        # instead of time-sharing it with the backward's own chip-filling layers.  The image is needed by the next step's Reconstructor only.
        self.tail_prefetch = getattr(TrainStep, 'tail_prefetch_default', True)
        self.tail_pause_res = getattr(TrainStep, 'tail_pause_res_default', None)      # None: the layers above min(output resolution / 2, 256)
        self.tail_hook_res = getattr(TrainStep, 'tail_hook_res_default', 16)
        # R's weight gradients likewise (single-GPU runs: with several ranks their all-reduce wants the whole backward to hide behind)
        self.wgrad_hook_res = getattr(TrainStep, 'wgrad_hook_res_default', 0)
        self.prepare_wt = getattr(TrainStep, 'prepare_wt_default', True)        # R's transposed weights at the start of the step, off the critical path
        self.mid_after_r = getattr(TrainStep, 'mid_after_r_default', False)      # development: the prefetched pass's middle stage behind R's backward instead of beside R
        self.torch_sampler = getattr(TrainStep, 'torch_sampler_default', False)       # sample() through torch's generator instead of wgs_sample_step
        self._draws = 0                      # batches drawn so far: the counter half of the sampler's Philox key
        self._sw = None                      # reconstructor.StepWeights of this engine (built with the first multi-stream step)
        self._pre = None                     # (z, idx, mag, img) drawn and generated one step ahead
        self._cold = True                    # next step builds the generator's weight caches of this arithmetic: single stream
        self._r_precision = r_precision
        self.steps_done = 0
        self.sampler_seed = sampler_seed(seed, rank, world, start_iter, device)
        self.gen.manual_seed(self.sampler_seed)
        r_params = [p for n, p in reconstructor.named_parameters()
                    if p.requires_grad and not n.startswith('features_extractor.fc')]
        s_params = [p for p in support_sets.parameters() if p.requires_grad]
        self.bucket = FlatBucket([(params.reconstructor_lr, r_params), (params.support_set_lr, s_params)], device)
        if world > 1:
            # replicas start from rank 0's parameters and BN running statistics, whatever each rank's RNG produced
            dist.broadcast(self.bucket.flat, 0)
            for b in reconstructor.buffers():
                if b.is_floating_point():
                    dist.broadcast(b.data, 0)
        K, n2 = support_sets.ALPHAS.shape
        self.K, self.n2, self.d = K, n2, support_sets.support_vectors_dim
        # the RBF kernels stream 16-byte vectors: a latent dimension that is not a multiple of 4 (BigGAN-256 / -512 truncate
        # dim_z to 119 / 112) runs on zero-padded copies of the table, the codes and the gradients (a few MB per step)
        self.dp = (self.d + 3) & ~3
        self.rbf_ws = rbf_workspace(local_batch, n2, self.dp, device)
        self.stats = torch.zeros(4, device=device)        # (ce, l1, total, accuracy) of the last step
        self.stats_sum = torch.zeros(4, device=device)
        self.stats_n = 0
        self.dlogits = torch.empty(local_batch, K, device=device)
        self.dmag = torch.empty(local_batch, device=device)
        self.argmax = torch.empty(local_batch, dtype=torch.int64, device=device)
        self.loss_ws = torch.empty(2 * local_batch, device=device)
        self.w_space = bool(getattr(params, 'shift_in_w_space', False))
        # 'mixed-strict' (what 'auto' resolves to for StyleGAN2): the per-layer table this engine calibrated on its generator's weights.  It
        # belongs to the ENGINE and travels with every generator call it makes (G(..., policy=)): the generator object is not touched, so
        # other engines / callers sharing it keep their own arithmetic.
        self.strict_calibration, self._policy = None, None
        self.calibrate_images = calibrate_images if calibrate_images is not None else getattr(TrainStep, 'calibrate_images_default', None)   # None: conv.STRICT_IMAGES
        self._checks = 0                     # run-time precision checks done so far (the check's sample is a function of this count only)
        if self.precision == C.MIXED_STRICT:
            self.calibrate_strict()
        self.comm_events = None      # set to a list to collect (start, end) HIP events around the all-reduce waits
        self.allreduce_bytes = 4 * self.bucket.flat.numel()     # payload of the step's collectives (R group + S group)

    # -- optimizer state (optional 'optim' key of checkpoint.pt; reference readers ignore unknown keys) ---------
    def optim_state(self):
        b = self.bucket
        return {'step': b.step_count, 'exp_avg': b.exp_avg.detach().cpu().clone(), 'exp_avg_sq': b.exp_avg_sq.detach().cpu().clone(),
                'layout': [(n, off) for (_, off, n) in b.segments]}

    def load_optim_state(self, st):
        b = self.bucket
        if st is None or st.get('layout') != [(n, off) for (_, off, n) in b.segments]:
            return False
        b.step_count = int(st['step'])
        b.exp_avg.copy_(st['exp_avg'].to(b.exp_avg.device))
        b.exp_avg_sq.copy_(st['exp_avg_sq'].to(b.exp_avg_sq.device))
        return True

    # -- arithmetic of this engine ---------------------------------------------------------------------
    def set_precision(self, precision, r_precision=None):
        """Switch this engine's generator (and, under 'auto', reconstructor) arithmetic; a batch generated ahead in the old mode
        is dropped and the next step runs single-stream (it builds the new mode's weight planes)."""
        from .reconstructor import r_arith
        self.precision = self.G.resolve_precision(precision)
        if r_precision is not None:
            self._r_precision = r_precision
        self.r_arith = r_arith(self._r_precision, self.precision)
        self._pre, self._cold = None, True
        self._policy, self.strict_calibration = None, None
        if self.precision == C.MIXED_STRICT:
            self.calibrate_strict()

    def _gkw(self):
        """Extra keyword arguments of this engine's generator calls: the calibrated per-layer table, when there is one."""
        return {'policy': self._policy} if self._policy is not None else {}

    def check_precision(self, gate=1e-3, batches=1):
        """Run-time guard of a 16-bit generator mode (the per-architecture policy tables were calibrated on random-init weights):
        image error of this engine's arithmetic against the exact-fp32 kernels on `batches` fresh batches of latent codes, with THIS
        generator's weights.  Max-norm relative error, of each batch tensor (how the parity tests apply the north_star's 1e-3 gate)
        and per single image.  Draws from its own generator (identical on every rank): the training sample stream is untouched.  None for fp32 engines."""
        if not C.is_reduced(self.precision):
            return None
        g = torch.Generator(device=self.dev)
        # the same codes on every rank (every rank takes the same decision), and a function of the number of checks made so far only: the
        # record of an engine's first check does not depend on how many steps ran before it (with `+ steps_done` the bench's
        # verdict flipped with --steps)
        g.manual_seed(0x5DEECE66D + self._checks)
        self._checks += 1
        per, bat = [], []
        for _ in range(max(1, int(batches))):
            z = sample_z(self.B, self.G.dim_z, truncation=getattr(self.p, 'z_truncation', None), device=self.dev, generator=g)
            # a class-conditional wrapper draws its class ids per call (numpy RNG, different on every rank): both passes must render the
            # SAME classes, and every rank the same ones — drawn here from the check's own generator
            kw = {}
            tc = getattr(self.G, 'target_classes', None)
            if tc is not None and hasattr(self.G, 'mixed_classes'):
                pool = tc.detach().reshape(-1).to(self.dev)
                kw['classes'] = pool[torch.randint(0, pool.numel(), (self.B,), device=self.dev, generator=g)]
            with torch.no_grad():
                ref = self.G(z, precision='fp32', **kw)
                img = self.G(z, precision=self.precision, **kw, **self._gkw())
                d, m = (img - ref).abs().flatten(1).amax(1), ref.abs().flatten(1).amax(1)
                per.append(d / m.clamp_min(1e-30))
                bat.append(d.max() / m.max().clamp_min(1e-30))
                del ref, img
        per, bat = torch.cat(per).cpu(), torch.stack(bat).cpu()
        batch = float(bat.max())
        self._cold = True       # the fp32 pass may have built weight caches on this stream: keep the next step single-stream
        return {'precision': C.precision_name(self.precision), 'batch': batch, 'batch_median': float(bat.median()),
                'per_image_median': float(per.median()), 'per_image_p99': float(per.quantile(0.99)), 'per_image_max': float(per.max()),
                'over_gate_frac': float((per > gate).float().mean()), 'gate': gate, 'ok': bool(batch < gate and batch == batch),
                'policy': (self.strict_calibration or {}).get('table'),
                'n': int(per.numel())}

    def calibrate_strict(self, images=None, margin=0.95, gate=1e-3, seed=0x51C7, ref_cache_bytes=8 << 30):
        """'mixed-strict': measure conv.STRICT_LADDER on THIS generator's weights (per-image max-norm error against the exact-fp32 kernels over
        `images` latent codes, identical on every rank) and adopt the first — cheapest — table whose worst single image is under
        margin * gate.  The table is kept on the engine (self._policy) and passed with every generator call; the generator object is not
        modified.  The fp32 reference images are kept on the device only while they fit `ref_cache_bytes` (StyleGAN2-1024: 576 images are
        7 GB, 2 304 would be 29 GB) — beyond that each rung recomputes them batch by batch; a rung stops at its first batch over the limit.  With several
        ranks, rank 0's choice is broadcast.  Returns the calibration record (also kept as self.strict_calibration)."""
        inner = getattr(self.G, 'G', None)
        size = getattr(inner, 'size', None)
        ladder = C.STRICT_LADDER.get(size)
        if self.precision != C.MIXED_STRICT or ladder is None or not hasattr(inner, 'mixed_policy'):
            return None
        images = images if images is not None else self.calibrate_images
        images = int(images if images is not None else C.STRICT_IMAGES.get(size, 576))
        nb = max(1, images // self.B)
        g = torch.Generator(device=self.dev)
        g.manual_seed(seed)
        zs = [sample_z(self.B, self.G.dim_z, truncation=getattr(self.p, 'z_truncation', None), device=self.dev, generator=g) for _ in range(nb)]
        keep = nb * self.B * 3 * size * size * 4 <= ref_cache_bytes
        refs = {}

        def ref_of(k):
            if k in refs:
                return refs[k]
            r = self.G(zs[k], precision='fp32')
            r = (r, r.abs().flatten(1).amax(1).clamp_min(1e-30))
            if keep:
                refs[k] = r
            return r
        tried, chosen = [], None
        limit = margin * gate
        with torch.no_grad():
            for ri, (name, pol) in enumerate(ladder):
                worst, seen, pend = 0.0, 0, None
                for k in range(nb):
                    r, m = ref_of(k)
                    e = ((self.G(zs[k], precision=self.precision, policy=pol) - r).abs().flatten(1).amax(1) / m).max()
                    pend = e if pend is None else torch.maximum(pend, e)
                    seen += 1
                    if k % 8 == 7 or k == nb - 1:                 # one host sync per eight batches; a rung stops at its first group over the limit
                        v, pend = float(pend), None
                        worst = v if v != v else max(worst, v)
                        if not (worst < limit):
                            break
                tried.append((name, worst, seen * self.B))
                if worst == worst and worst < limit:
                    chosen = ri
                    break
        if chosen is None:       # (cannot happen with a split-bf16 last rung unless the generator overflows: keep it anyway)
            chosen = len(ladder) - 1
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            t = torch.tensor([chosen], dtype=torch.int64, device=self.dev if dist.get_backend() == 'nccl' else 'cpu')
            dist.broadcast(t, 0)
            chosen = int(t.item())
        self._policy = ladder[chosen][1]
        self._pre, self._cold = None, True
        self.strict_calibration = {'table': ladder[chosen][0], 'tried': [(n, float('%.3g' % w), ni) for n, w, ni in tried], 'images': nb * self.B,
                                   'margin': margin, 'gate': gate, 'fp16_layers': self._policy.spends()}
        return self.strict_calibration

    def _out_size(self):
        """Output resolution of the generator (StyleGAN2: .size; ProgGAN: from its block count); 256 when it cannot be told."""
        inner = getattr(self.G, 'G', None)
        if hasattr(inner, 'size'):
            return int(inner.size)
        if hasattr(inner, 'num_blocks'):
            return 4 << ((inner.num_blocks - 2) // 2)
        if hasattr(inner, 'resolution'):
            return int(inner.resolution)
        return 256

    # -- sampling (lib/trainer.py:195-221), on the device ---------------------------------------------
    def sample(self):
        """(z [B, d], path indices [B] int64, shift magnitudes [B]) of one step, drawn in HBM by ONE launch (wgs_sample_step: the same
        distributions as lib/trainer.py:195-221 incl. the arange-weighted draw without replacement; Philox stream keyed by this rank's
        sampler seed and the draw counter).  `torch_sampler` (development / tests): the same draws from torch's generator, ~25 launches."""
        p, B = self.p, self.B
        if self.torch_sampler or B > 1024:
            return self._sample_torch()
        z = torch.empty(B, self.G.dim_z, device=self.dev)
        idx = torch.empty(B, dtype=torch.int64, device=self.dev)
        mag = torch.empty(B, device=self.dev)
        trunc = getattr(p, 'z_truncation', None)
        L.check(L.lib().wgs_sample_step(L.ptr(z), L.ptr(idx, torch.int64), L.ptr(mag), B, self.G.dim_z, self.K, L.c_float(p.min_shift_magnitude),
                                        L.c_float(p.max_shift_magnitude), L.c_float(0.0 if trunc is None else float(trunc)),
                                        ctypes.c_uint64(self.sampler_seed & 0xFFFFFFFFFFFFFFFF), ctypes.c_uint64(self._draws), L.stream()), 'wgs_sample_step')
        self._draws += 1
        return z, idx, mag

    def _sample_torch(self):
        p, B = self.p, self.B
        z = sample_z(B, self.G.dim_z, truncation=getattr(p, 'z_truncation', None), device=self.dev, generator=self.gen)
        idx = torch.randint(0, self.K, (B,), device=self.dev, generator=self.gen)
        lo, hi = p.min_shift_magnitude, p.max_shift_magnitude
        pos = (lo - hi) * torch.rand(B, device=self.dev, generator=self.gen) + hi
        neg = (lo - hi) * torch.rand(B, device=self.dev, generator=self.gen) - lo
        pool = torch.cat((neg, pos))
        ids = torch.arange(2 * B, dtype=torch.float, device=self.dev)          # weights 0..2B-1, :218
        mag = pool[torch.multinomial(ids, B, replacement=False, generator=self.gen)]
        return z, idx, mag

    def step(self, z=None, idx=None, mag=None):
        """One training step.  Enqueues on the engine's high-priority stream (ordered after the caller's current stream, which in turn
        waits for the step: the caller sees ordinary stream semantics) or, without it, on the current stream."""
        if self.main_stream is None or not self.two_streams:
            return self._step(z, idx, mag)
        outer = torch.cuda.current_stream(self.dev)
        self.main_stream.wait_stream(outer)
        with torch.cuda.stream(self.main_stream):
            st = self._step(z, idx, mag)
        outer.wait_stream(self.main_stream)
        return st

    def _step(self, z=None, idx=None, mag=None):
        G, S, R, p, B = self.G, self.S, self.R, self.p, self.B
        lib, st = L.lib(), L.stream()
        auto = z is None
        img = None
        pre_ev = None
        if auto:
            if self._pre is not None:
                z, idx, mag, img, pre_ev = self._pre
                self._pre = None
            else:
                z, idx, mag = self.sample()
        self.bucket.zero_grad()
        # The un-shifted pass G(z) (nothing saved, :200) runs on a side stream next to the shifted pass: both are the same
        # network on independent inputs, and their 4x4 .. 16x16 layers each fill only part of the chip.
        cur = torch.cuda.current_stream(self.dev)
        # (not in the very first step: the generator builds its packed / split weight caches lazily in its first forward, and
        # those must be produced on the main stream, ahead of everything that reads them)
        side = self.side_stream if (self.two_streams and not self._cold) else None
        self._cold = False
        prec, gkw = self.precision, self._gkw()
        pre_img = img is not None           # G(z) of this batch was generated during the previous step (see below)
        if pre_img:
            img.record_stream(cur)          # (the wait for it sits in front of the Reconstructor, its only reader)
        elif side is not None:
            side.wait_stream(cur)
            with torch.cuda.stream(side), torch.no_grad():
                img = G(z, precision=prec, **gkw)
        with torch.no_grad():
            if img is None:
                img = G(z, precision=prec, **gkw)                                             # :200, nothing saved
            code = G.get_w(z) if self.w_space else z                          # :236
        # Derived forms of R's trained weights (transposed copies for its input-gradient convs, Winograd operands in 'fp32w'): fixed since
        # the last Adam update, re-derived now on the side stream instead of launch by launch inside R's forward / backward
        sw, wt_ev = None, None
        if side is not None and self.prepare_wt and getattr(R, 'reconstructor_type', None) == 'ResNet':
            if self._sw is None:
                from .reconstructor import StepWeights
                self._sw = StepWeights(R)
            sw = self._sw
            side.wait_stream(cur)
            with torch.cuda.stream(side), torch.no_grad():
                sw.refresh()
            wt_ev = torch.cuda.Event()
            wt_ev.record(side)
        # G(z) has no trainable ancestor (:200), so the NEXT step's un-shifted pass does not depend on this step's update: its batch
        # is drawn now (same order of draws from the sampler's generator as one draw per step) and generated on a third stream.  Its
        # layers up to 32 x 32 — latency-bound: a tenth of the FLOPs, a quarter of a pass's time — are enqueued HERE, so that they run
        # next to the same layers of this batch's shifted pass (two latency-bound chains side by side); the chip-filling rest is
        # gated behind the shifted forward (below), where it fills the CUs that the Reconstructor's short launches leave idle.
        nxt = None
        static = getattr(self, 'debug_static_unshifted', False)       # development (tools/ab_tail.py): re-use ONE un-shifted batch for ever — the
        if static and auto and pre_img:                               # step without its side work, i.e. the critical chain's own time.  NOT training.
            self._pre = (z, idx, mag, img, None)
        elif auto and side is not None and self.prefetch:
            zn, idxn, magn = self.sample()
            handle = None
            if self.split_prefetch and hasattr(G, 'begin') and not self.w_space:
                self.pre_stream.wait_stream(cur)
                tail_res = self.tail_pause_res or min(self._out_size() // 2, 256)      # (1024^2 generators: 512 / 256 / 128 measured 32.7 / 32.3 / 32.3 ms at cfg5)
                pauses = (self.split_pause_res, tail_res) if (self.tail_prefetch and tail_res > self.split_pause_res) else self.split_pause_res
                with torch.cuda.stream(self.pre_stream), torch.no_grad():
                    handle = G.begin(zn, precision=prec, pause_res=pauses, **gkw)
                zn.record_stream(self.pre_stream)
            nxt = (zn, idxn, magn, handle)
        # shift = mag * S(mask, code)   (:235) — fused scale
        lg = S.LOGGAMMA.reshape(-1) if S.learn_gammas else None
        pad = self.dp - self.d
        if pad:
            table = torch.nn.functional.pad(S.SUPPORT_SETS.detach().view(self.K, self.n2, self.d), (0, pad)).reshape(self.K, -1).contiguous()
            code_k = torch.nn.functional.pad(code, (0, pad)).contiguous()
        else:
            table, code_k = S.SUPPORT_SETS, code
        shift = torch.empty(B, self.dp, device=self.dev)
        L.check(lib.wgs_rbf_fwd(L.ptr(table), L.ptr(S.ALPHAS), L.ptr(lg), L.c_float(float(S.gamma)),
                                L.ptr(idx, torch.int64), L.ptr(code_k), L.ptr(mag), L.ptr(shift), L.ptr(self.rbf_ws),
                                B, self.K, self.n2, self.dp, st), 'wgs_rbf_fwd')
        if pad:
            shift = shift[:, :self.d].contiguous()
        shift.requires_grad_(True)
        # side work to be enqueued from inside THIS forward's backward (the prefetched pass's tail, deferred weight gradients): the list
        # is bound to the forward's autograd node by the generator (Generator.bwd_hooks) — entries are appended below, once they exist;
        # no other forward / backward of the generator sees them
        inner = getattr(G, 'G', None)
        hookable = hasattr(inner, 'bwd_hooks')
        hooks = []
        if hookable:
            inner.bwd_hooks = hooks
        try:
            img_shifted = G(z, shift, precision=prec, **gkw)                                  # :239, input-gradient only
        finally:
            if hookable:
                inner.bwd_hooks = None      # (taken by the forward; cleared here if it failed or kept nothing)
        if side is not None and not pre_img:
            cur.wait_stream(side)
            img.record_stream(cur)
        # the next batch's un-shifted pass (its remaining, chip-filling layers), gated behind this step's shifted forward.  Same
        # arithmetic, same values as an ordinary call; one generated batch stays unused when training stops.
        tail_box = [None]

        def run_mid():
            zn, idxn, magn, handle = nxt
            self.pre_stream.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(self.pre_stream), torch.no_grad():
                if handle is None:
                    imgn = G(zn, precision=prec, **gkw)
                elif self.tail_prefetch:
                    imgn = G.advance(handle)            # None: paused in front of the tail layers
                else:
                    imgn = G.finish(handle)
            zn.record_stream(self.pre_stream)
            if imgn is not None:
                ev = torch.cuda.Event()
                ev.record(self.pre_stream)
                self._pre = (zn, idxn, magn, imgn, ev)
            else:
                def tail():
                    # called from the generator's backward (autograd's device thread: its current stream is the step's stream)
                    self.pre_stream.wait_stream(torch.cuda.current_stream(self.dev))
                    with torch.cuda.stream(self.pre_stream), torch.no_grad():
                        im = G.finish(handle)
                    ev = torch.cuda.Event()
                    ev.record(self.pre_stream)
                    self._pre = (zn, idxn, magn, im, ev)
                tail_box[0] = tail
        if nxt is not None and not self.mid_after_r:
            run_mid()
        if pre_img:                       # the image generated one step ahead: complete before the Reconstructor reads it
            if pre_ev is not None:
                cur.wait_event(pre_ev)
            else:
                cur.wait_stream(self.pre_stream)
        if wt_ev is not None:
            cur.wait_event(wt_ev)
        logits, mag_hat, saved = R._forward_impl(img, img_shifted.detach(), save=True, arith=self.r_arith, **({'sw': sw} if sw else {}))   # :242
        L.check(lib.wgs_ce_l1_loss(L.ptr(logits), L.ptr(idx, torch.int64), L.ptr(mag_hat.reshape(B)), L.ptr(mag),
                                   L.c_float(p.lambda_cls), L.c_float(p.lambda_reg), L.ptr(self.dlogits), L.ptr(self.dmag),
                                   L.ptr(self.stats), L.ptr(self.argmax, torch.int64), L.ptr(self.loss_ws), B, self.K, st),
                'wgs_ce_l1_loss')                                             # :245-249,257-258
        gb = self.bucket.gview
        # R's conv weight gradients are not needed for d_img: they are queued and run on the side stream, next to the
        # generator's backward (whose 4x4..32x32 layers under-fill the chip); the ResNet path only (LeNet computes them inline)
        deferred = None
        if side is not None and R.reconstructor_type == 'ResNet' and self.defer_wgrad:
            # (eager_wgrad: a layer's weight gradient goes to the side stream as soon as its dy exists, i.e. next to the REST of R's
            # backward instead of next to the generator's backward.  Measured slower — auto 26.47 -> 26.71 ms, fp32w +-0: during R's
            # phases the prefetched pass already fills the chip, a third stream only adds contention.  Off.)
            deferred = _EagerSide(side) if self.eager_wgrad else []
        _, _, d_img = R._backward_impl(saved, self.dlogits, self.dmag, need_x=(False, True), gbuf=gb, deferred=deferred, **({'sw': sw} if sw else {}))
        del saved
        if nxt is not None and self.mid_after_r:
            run_mid()
        tail = tail_box[0]
        pending = []

        def run_wgrads(deferred=deferred):
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                if not isinstance(deferred, _EagerSide):
                    for x_, dy_, fn in deferred:
                        fn()
                        x_.record_stream(side); dy_.record_stream(side)
                if self.world > 1:
                    # R's gradients (the first bucket group, 47 MB at cfg3) are final after these launches: their all-reduce
                    # is queued behind them and overlaps the generator's backward (no trainable parameters)
                    _, a, b = self.bucket.groups[0]
                    pending.append(dist.all_reduce(self.bucket.grad[a:b], async_op=True))
        if deferred is not None:
            if self.wgrad_hook_res and hookable and self.world == 1:
                hooks.append((self.wgrad_hook_res, run_wgrads))     # under the backward's latency-bound tail instead of next to its chip-filling layers
            else:
                run_wgrads()
            del deferred
        elif self.world > 1:
            _, a, b = self.bucket.groups[0]
            pending.append(dist.all_reduce(self.bucket.grad[a:b], async_op=True))
        if tail is not None:
            hooks.append((self.tail_hook_res, tail))      # (a generator without hooks: run after its backward, below)
        try:
            img_shifted.backward(d_img)                                       # G: d image -> d shift
        finally:
            left, hooks[:] = list(hooks), []    # consumed by the synthesis backward (it empties the list it was handed); what is left did not
        for h in sorted(left, key=lambda q: -q[0]):                          # pass through it (no hook support, a backward that kept nothing): run now
            h[1]()
        dtable = gb[id(S.SUPPORT_SETS)]
        dlg = gb[id(S.LOGGAMMA)].reshape(-1) if (S.learn_gammas and id(S.LOGGAMMA) in gb) else None
        dal = gb[id(S.ALPHAS)] if (S.learn_alphas and id(S.ALPHAS) in gb) else None
        gshift = shift.grad.contiguous()
        if pad:
            gshift = torch.nn.functional.pad(gshift, (0, pad)).contiguous()
            dtable_k = torch.zeros(self.K, self.n2 * self.dp, device=self.dev)
        else:
            dtable_k = dtable
        L.check(lib.wgs_rbf_bwd(L.ptr(table), L.ptr(S.ALPHAS), L.ptr(lg), L.c_float(float(S.gamma)),
                                L.ptr(idx, torch.int64), L.ptr(code_k), L.ptr(mag), L.ptr(gshift), L.ptr(self.rbf_ws),
                                L.ptr(dtable_k), L.ptr(dlg), L.ptr(dal), None, B, self.K, self.n2, self.dp, st), 'wgs_rbf_bwd')
        if pad:
            dtable.view(self.K, self.n2, self.d).copy_(dtable_k.view(self.K, self.n2, self.dp)[:, :, :self.d])
        if side is not None:
            cur.wait_stream(side)                                             # deferred weight gradients (and their all-reduce)
        if self.world > 1:
            _, a, b = self.bucket.groups[1]                                   # S's gradients (RCCL sum; Adam divides by world)
            pending.append(dist.all_reduce(self.bucket.grad[a:b], async_op=True))
            ev = self.comm_events
            if ev is not None:      # bench.py: how long the main stream sits in front of the collectives (exposed wait)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            for w in pending:
                w.wait()
            if ev is not None:
                e1.record()
                ev.append((e0, e1))
        self.bucket.adam_step(world=self.world)                               # :253-254
        self.stats_sum += self.stats
        self.stats_n += 1
        self.steps_done += 1
        return self.stats

    def pop_stats(self):
        """Mean (accuracy, classification_loss, regression_loss, total_loss) since the last call (one sync)."""
        s = self.stats_sum.clone()
        if self.world > 1:
            dist.all_reduce(s)
            s /= self.world
        v = (s / max(self.stats_n, 1)).tolist()
        self.stats_sum.zero_()
        self.stats_n = 0
        return {'accuracy': v[3], 'classification_loss': v[0], 'regression_loss': v[1], 'total_loss': v[2]}


class Trainer(object):
    def __init__(self, params=None, exp_dir=None, use_cuda=False, multi_gpu=False, root="experiments"):
        if params is None:
            raise ValueError("Cannot build a Trainer instance with empty params: params={}".format(params))
        self.params = params
        self.use_cuda = use_cuda
        self.multi_gpu = multi_gpu
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.tensorboard = getattr(params, 'tensorboard', False)
        self.wip_dir = osp.join(root, "wip", exp_dir)
        self.complete_dir = osp.join(root, "complete", exp_dir)
        self.stats_json = osp.join(self.wip_dir, 'stats.json')
        self.models_dir = osp.join(self.wip_dir, 'models')
        self.checkpoint = osp.join(self.models_dir, 'checkpoint.pt')
        if self.rank == 0:
            os.makedirs(self.models_dir, exist_ok=True)
            if not osp.isfile(self.stats_json):
                with open(self.stats_json, 'w') as out:
                    json.dump({}, out)
        self.tb_writer = None
        if self.tensorboard and self.rank == 0:
            try:   # optional dependency (absent in this image); the reference imports it unconditionally
                from torch.utils.tensorboard import SummaryWriter
                self.tb_dir = osp.join(self.wip_dir, 'tensorboard')
                os.makedirs(self.tb_dir, exist_ok=True)
                self.tb_writer = SummaryWriter(log_dir=self.tb_dir)
            except Exception as e:  # noqa: BLE001
                print("#. TensorBoard unavailable ({}); continuing without it".format(e))
        # Whether statistics are popped EVERY iteration (TensorBoard scalars) — a collective when world > 1 (TrainStep.pop_stats
        # all-reduces), so the decision must be the same on every rank although only rank 0 owns a writer: agreed on in train().
        self.tb_active = self.tb_writer is not None
        self.iter_times = np.array([])
        self.stat_tracker = TrainingStatTracker()

    @staticmethod
    def _plain(sd):
        return {k: v.detach().clone(memory_format=torch.contiguous_format) for k, v in sd.items()}

    def get_starting_iteration(self, support_sets, reconstructor):
        """Resume from models/checkpoint.pt if present (lib/trainer.py:74-89): {'iter','support_sets','reconstructor'}."""
        starting_iter = 1
        self.resume_optim = None
        self.ck_iter = None
        if osp.isfile(self.checkpoint):
            ck = torch.load(self.checkpoint, map_location='cpu')
            starting_iter = self.ck_iter = ck['iter']
            support_sets.load_state_dict(ck['support_sets'])
            reconstructor.load_state_dict(ck['reconstructor'])
            # extension: Adam moments (the reference restarts its optimizers from zero on resume, lib/trainer.py:153-156,
            # 288-295 stores only the three keys above; its readers index by key, so the extra key is ignored there).
            # A reference-layout checkpoint resumes as the reference does: iteration ck['iter'] is run again with fresh
            # optimizers.  With the moments present the state is exactly the one AFTER iteration ck['iter'] (parameters,
            # moments, Adam step count), so the run continues at ck['iter'] + 1 — no step is applied twice and the
            # bias-correction count stays in step with the iteration number.
            self.resume_optim = ck.get('optim')
            if self.resume_optim is not None:
                starting_iter = ck['iter'] + 1
        return starting_iter

    def log_progress(self, iteration, mean_iter_time, elapsed_time, eta, stats):
        with open(self.stats_json) as f:
            stats_dict = json.load(f)
        stats_dict.update({iteration: stats})
        with open(self.stats_json, 'w') as out:
            json.dump(stats_dict, out)
        update_progress("  \\__.Training [bs: {}] [iter: {:06d}/{:06d}] ".format(
            self.params.batch_size, iteration, self.params.max_iter), self.params.max_iter, iteration + 1)
        if iteration < self.params.max_iter - 1:
            print()
        print("      \\__Batch accuracy      : {:.03f}".format(stats['accuracy']))
        print("      \\__Classification loss : {:.08f}".format(stats['classification_loss']))
        print("      \\__Regression loss     : {:.08f}".format(stats['regression_loss']))
        print("      \\__Total loss          : {:.08f}".format(stats['total_loss']))
        print("         ===================================================================")
        print("      \\__Mean iter time      : {:.3f} sec".format(mean_iter_time))
        print("      \\__Elapsed time        : {}".format(sec2dhms(elapsed_time)))
        print("      \\__ETA                 : {}".format(sec2dhms(eta)))
        print("         ===================================================================")
        if sys.stdout.isatty():
            update_stdout(10)

    def agree_tensorboard(self, device=None):
        """Rank 0's "a TensorBoard writer exists" broadcast to every rank (the writer is rank 0's alone, and creating it can fail there:
        the other ranks cannot derive it from --tensorboard).  pop_stats() contains an all-reduce: every rank must enter it in the
        same iterations."""
        flag = 1 if self.tb_writer is not None else 0
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            on_dev = dist.get_backend() == 'nccl' and device is not None
            t = torch.tensor([flag], dtype=torch.int64, device=device if on_dev else 'cpu')
            dist.broadcast(t, 0)
            flag = int(t.item())
        self.tb_active = bool(flag)
        return self.tb_active

    def stats_due(self, iteration):
        """True in the iterations whose statistics are popped (one device sync + one all-reduce when world > 1): a function of
        rank-invariant state only."""
        return bool(self.tb_active or iteration % self.params.log_freq == 0)

    def precision_check(self, engine, iteration):
        """--check-precision: measure the generator arithmetic's image error on the current weights; a mode that misses the
        1e-3 gate is replaced by the fp32-class split-bf16 for the rest of the run (every rank measures the same codes with the
        same frozen generator, so every rank takes the same decision)."""
        r = engine.check_precision()
        if r is None:
            return None
        if self.rank == 0:
            print("#. Precision check @ iteration {}: {} image error vs exact fp32: batch {:.2e}, single image median {:.2e} / max {:.2e}, {:.2%} over "
                  "(gate {:.0e}, {} codes) -- {}".format(iteration, r['precision'], r['batch'], r['per_image_median'], r['per_image_max'],
                                                          r['over_gate_frac'], r['gate'], r['n'], 'ok' if r['ok'] else 'OVER THE GATE'))
        fb = C.AUTO_FALLBACK_BY_FAMILY['stylegan2'] if hasattr(getattr(engine.G, 'G', None), 'mixed_policy') else C.AUTO_FALLBACK
        if not r['ok'] and engine.precision not in (C.PRECISION_NAMES[C.AUTO_FALLBACK], C.PRECISION_NAMES[fb]):
            engine.set_precision(fb)
            if self.rank == 0:
                print("#. Generator arithmetic switched to {} (fp32-class) for the rest of the run".format(fb))
        return r

    def train(self, generator, support_sets, reconstructor):
        p = self.params
        if not (self.use_cuda and torch.cuda.is_available()):
            raise L.WgsError("this Trainer drives the HIP kernels: a GPU is required (the reference's CPU path is "
                             "restated only in oracle/, for checking)")
        dev = torch.device('cuda', torch.cuda.current_device())
        if self.rank == 0:
            torch.save(self._plain(support_sets.state_dict()), osp.join(self.models_dir, 'support_sets_init.pt'))
        generator.to(dev).eval()
        support_sets.to(dev).train()
        reconstructor.to(dev).train()
        starting_iter = self.get_starting_iteration(support_sets, reconstructor)
        if self.world > 1:   # identical replicas on every rank
            for t in list(support_sets.parameters()) + list(reconstructor.parameters()) + list(reconstructor.buffers()):
                dist.broadcast(t.data, 0)
        if self.ck_iter == p.max_iter:
            print("#. This experiment has already been completed and can be found @ {}".format(self.wip_dir))
            if self.rank == 0:
                try:
                    shutil.copytree(src=self.wip_dir, dst=self.complete_dir, ignore=shutil.ignore_patterns('checkpoint.pt'))
                except IOError as e:
                    print("  \\__Already exists -- {}".format(e))
            sys.exit()
        if p.batch_size % self.world:
            raise ValueError("--batch-size ({}) is the GLOBAL batch and must divide by the world size ({})".format(
                p.batch_size, self.world))
        engine = TrainStep(generator, support_sets, reconstructor, p, p.batch_size // self.world, dev, world=self.world,
                           seed=getattr(p, 'seed', None), rank=self.rank, start_iter=starting_iter,
                           precision=getattr(p, 'precision', None), r_precision=getattr(p, 'r_precision', 'auto'))
        if engine.strict_calibration and self.rank == 0:
            c = engine.strict_calibration
            print("#. Generator arithmetic calibrated on this generator ({} latent codes against the exact-fp32 kernels): per-layer table '{}' "
                  "({} fp16-rounded layer(s)); rungs tried (worst single image, codes seen): {}".format(
                      c['images'], c['table'], c['fp16_layers'], ', '.join("%s %.2e (%d)" % tuple(t) for t in c['tried'])))
        if getattr(self, 'resume_optim', None) is not None and engine.load_optim_state(self.resume_optim) and self.rank == 0:
            print("#. Restored Adam moments (step {}) from the checkpoint".format(engine.bucket.step_count))
        if self.rank == 0:
            print("#. Start training from iteration {}".format(starting_iter))
        self.agree_tensorboard(dev)
        check_every = int(getattr(p, 'check_precision', 0) or 0)
        if C.is_f16_operand(engine.precision):      # an fp16-operand mode: check it once against the exact-fp32 kernels on THESE weights
            self.precision_check(engine, starting_iter - 1)
        t0 = time.time()
        # Iteration time: the launches are asynchronous, so a per-iteration host clock would measure launch latency.  The
        # device is synchronised only where statistics are popped (log boundaries): time between those, per iteration.
        mark_t, mark_iter = t0, starting_iter - 1
        for iteration in range(starting_iter, p.max_iter + 1):
            engine.step()
            if check_every and C.is_reduced(engine.precision) and iteration % check_every == 0:
                self.precision_check(engine, iteration)
            if self.stats_due(iteration):
                stats = engine.pop_stats()           # one device sync (+ one all-reduce: every rank takes this branch in the same iterations)
                now = time.time()
                self.iter_times = np.append(self.iter_times, np.full(iteration - mark_iter, (now - mark_t) / (iteration - mark_iter)))
                mark_t, mark_iter = now, iteration
                if self.tb_writer is not None:
                    for key, value in stats.items():
                        self.tb_writer.add_scalar(key, value, iteration)
            elapsed_time = time.time() - t0
            eta = elapsed_time * ((p.max_iter - iteration) / (iteration - starting_iter + 1))
            if iteration % p.log_freq == 0 and self.rank == 0:
                self.log_progress(iteration, self.iter_times.mean(), elapsed_time, eta, stats)
            if iteration % p.ckp_freq == 0 and self.rank == 0:
                torch.save({'iter': iteration, 'support_sets': self._plain(support_sets.state_dict()),
                            'reconstructor': self._plain(reconstructor.state_dict()),
                            'optim': engine.optim_state()}, self.checkpoint)
        elapsed_time = time.time() - t0
        if self.rank == 0:
            torch.save(self._plain(support_sets.state_dict()), osp.join(self.models_dir, 'support_sets.pt'))
            torch.save(self._plain(reconstructor.state_dict()), osp.join(self.models_dir, 'reconstructor.pt'))
            print("#.Training completed -- Total elapsed time: {}.".format(sec2dhms(elapsed_time)))
            try:
                shutil.copytree(src=self.wip_dir, dst=self.complete_dir, ignore=shutil.ignore_patterns('checkpoint.pt'))
            except IOError as e:
                print("  \\__Already exists -- {}".format(e))
        return engine
