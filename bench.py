#!/usr/bin/env python
"""bench.py — training-step images/sec (warp -> G -> R -> loss -> backward -> Adam) on MI355X.

Headline workload (BASELINE.json `metric`, configs[2]): StyleGAN2-FFHQ-256 architecture (random-init weights drawn exactly
as the reference constructors do; no checkpoints offline), K=128 warping functions x N=32 dipoles, ResNet-18
reconstructor, batch 32 per GPU, Z-space shifts, --learn-gammas, synthetic z ~ N(0, I) sampled in HBM.

  python bench.py --gpus N --steps K --warmup W
The headline (`value`, `dtype`, `roofline`) is the REFERENCE's arithmetic: fp32 everywhere (f32-input MFMA, fp32 accumulate; the
reference computes in fp32), with the 3x3 stride-1 convolutions in the Winograd F(2x2,3x3) form
(`--precision fp32w`): fp32 operands, fp32 transforms, fp32 accumulate, 3e-6 against fp64.  The same
workload is timed with the same K / W in the product's default arithmetic (`auto` = the mixed fp16 / split-bf16 per-layer policy
of DESIGN.md section 3.2, calibrated on the engine's own generator so that no single image of its sample is over the 1e-3 gate: `product` in the line) and in DIRECT-form exact fp32 (`fp32`: no Winograd anywhere: `direct_fp32`); both
run at every N.  Short runs of the other modes / BASELINE configs are N=1 only (`others`: name -> images/sec).

OUTPUT CONTRACT: the LAST stdout line is ONE JSON object of < 4 KB (the driver keeps ~8 KB of stdout): metric, value, unit, n_gpus,
steps, warmup, ms_per_step, dtype, config, roofline, cpu_baseline, comm, product, direct_fp32, others.  Everything else - per-symbol and
per-shape kernel tables, hbm_subpaths, host overheads, full records of every extra run - goes to `--extra-out` (default
gpurun_out/bench_extra.json; the line's `extra_file` names it).

With N > 1 and no torch.distributed environment, bench.py starts the N ranks ITSELF (torch.distributed.run, one rank per
GPU, rendezvous on 127.0.0.1) and relays rank 0's JSON line; under an external launcher (RANK / WORLD_SIZE set) it runs
as one rank of that job and checks that WORLD_SIZE == --gpus.  Gradients of R and S are all-reduced over RCCL once per
step; per-GPU batch is fixed (weak scaling); the generator is never communicated.  `--dist-backend gloo` (or
WGS_DIST_BACKEND=gloo) swaps RCCL for gloo and lets the ranks share one device: the N > 1 code path end to end on a 1-GPU box.

Fields (rank 0):
  value / ms_per_step  whole-job images/sec over exactly --steps timed steps (barrier + device sync on both sides, max over ranks)
  roofline             the DOMINANT KERNEL of the step by kernel symbol (largest summed HIP-event time over all its launch
                       shapes, 2 extra single-stream steps).  achieved = MFMA FLOPs the kernel EXECUTES for the work / its summed
                       duration (for a direct-form kernel that is the algorithmic 2 * pixels * Cout * Cin * taps; the Winograd kernel
                       executes 16/36 of it), frac = achieved / dense MFMA peak of its operand dtype (<= 1 by construction);
                       direct_equiv_TFLOPs = the algorithmic (direct-form) FLOPs / the same duration; traffic / traffic_algorithmic =
                       launch-weighted mean HBM bytes per launch over the symbol's shapes from the committed rocprofv3 --pmc passes
  cpu_baseline         the oracle's replay of the reference step as written on this box's host cores (bounded sample)
  comm                 (N > 1) backend, world size, all-reduce payload per step, exposed wait of the main stream per step
  side file            host (launches / enqueue time per step), hbm_subpaths (GB/s of the HBM-bound sub-kernels), by_symbol, kernel_shapes
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
import types

# multi-process GPU work on this driver needs dmabuf IPC (RCCL / hipIpc*): keep it set before HIP initialises
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch
import torch.distributed as dist

# SURVEY.md section 8(d) / BASELINE.md section 3: algorithmic GFLOP per training image (G fwd x2 + G dgrad + R fwd + R bwd)
GFLOP_PER_IMG = {'stylegan2-256': 285.8, 'stylegan2-1024': 687.8, 'proggan-1024': 498.7, 'proggan-256': 184.3, 'biggan-128': 127.5}
FP32_MFMA_PEAK_TF = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
F16_MFMA_PEAK_TF = 2500.0      # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_{bf16,f16}, dense (no sparsity)
MFMA_PER_PRODUCT = {'fp32': 1.0, 'fp32w': 16.0 / 36.0, 'bf16x3': 3.0, 'bf16x3w': 2.0, 'f16': 1.0, 'f16x2': 2.0}        # per launch label (a 'mixed' run has all three 16-bit kinds)
DTYPE = {'fp32': 'fp32', 'fp32w': 'fp32', 'bf16x3': 'bf16x3 (split-bf16, fp32-class)', 'f16': 'fp16 operands, fp32 accumulate',
         'f16x2': 'fp16 x2 operands, fp32 accumulate', 'mixed': 'mixed fp16 / split-bf16 per layer, fp32 accumulate',
         'mixed-strict': 'mixed fp16 x2 / split-bf16 per layer (strict table), fp32 accumulate',
         'bf16x3w': 'bf16x3 (split-bf16, fp32-class), F(2,3) form of the 3x3 stride-1 convs'}
DTYPE_TEXT = {
    'fp32': "fp32 everywhere (the reference's arithmetic): every conv of G and R, forward and backward, is f32-input MFMA "
            "(v_mfma_f32_32x32x2_f32) with fp32 accumulate; everything else fp32 VALU",
    'fp32w': "fp32 everywhere, as 'fp32', with the 3x3 stride-1 convs of G and R (forward and input-gradient) in the Winograd F(2x2,3x3) "
             "form on the f32-input MFMA (16 instead of 36 multiplies per 2x2 outputs, transforms in fp32 inside the kernel; 3e-6 against "
             "fp64 convolutions, no wider than the direct fp32 kernel)",
    'bf16x3': "bf16x3: generator convs split every fp32 operand into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate (~2^-16)",
    'f16': "f16: generator convs round operands to fp16 (dynamic power-of-two scale on every operand), 1 fp16 MFMA per product, "
           "fp32 accumulate / demodulation / epilogue (image error vs the fp32 kernels ~8e-4 at 256^2: on the 1e-3 gate, reported only)",
    'f16x2': "f16x2: as f16 with the frozen weights as fp16 hi+lo, 2 fp16 MFMAs per product",
    'mixed': "mixed fp16: per-layer arithmetic of the generator by an image-error budget (gate 1e-3; conv.MixedPolicy, DESIGN.md section "
             "3.2): fp16 operands (1 or 2 MFMAs per product) in the layers at >= 64x64, split-bf16 x3 (fp32-class) below; fp32 accumulate / "
             "demodulation / epilogue everywhere; dynamic power-of-two scale on every fp16 operand",
}
DTYPE_TEXT['bf16x3w'] = DTYPE_TEXT['bf16x3'] + "; the 3x3 stride-1 convs with their horizontal taps in the Winograd form F(2,3) (V and U split into bf16 hi+lo after the fp32 transforms: 2 MFMAs per direct product)"
DTYPE_TEXT['mixed-strict'] = DTYPE_TEXT['mixed'] + " — per-layer table CALIBRATED on the engine's own generator (conv.STRICT_LADDER): the cheapest rung with no single image of a 2 304-code sample over 0.95e-3"
R_TEXT = {(5, 5, 0): "; reconstructor (trained): fp32 MFMA, Winograd form of the 3x3 stride-1 forward / input-gradient convs, direct exact "
                     "fp32 for the rest and for every weight gradient; BatchNorm statistics in fp64 partials",
          (0, 0, 0): "; reconstructor (trained): exact fp32 MFMA forward, input-gradient and weight-gradient convs; BatchNorm statistics in fp64 partials",
          (1, 1, 1): "; reconstructor (trained): fp32-class only - split-bf16 x3 (3 MFMAs, ~2^-16 per product) forward, input-gradient and "
                     "the wide weight-gradient convs, exact fp32 MFMA for the other weight gradients; BatchNorm statistics in fp64 partials"}


def make_params(w_space=False):
    return types.SimpleNamespace(reconstructor_lr=1e-4, support_set_lr=1e-4, min_shift_magnitude=0.25,
                                 max_shift_magnitude=0.45, lambda_cls=1.0, lambda_reg=0.25, z_truncation=None,
                                 shift_in_w_space=w_space)


ENGINE_KW = {}      # --single-stream: TrainStep(two_streams=False), so that a rocprofv3 kernel table adds up to the step


def build(dev, gan, K, N, B, rank=0, w_space=False, size=256, precision='fp32', r_precision='auto', world=None):
    """gan: 'stylegan2' (size 256 / 1024), 'proggan' (size 1024 native or 256 = first 14 blocks), 'biggan' (size 128 / 256)."""
    from warpedganspace_amd.reconstructor import Reconstructor
    from warpedganspace_amd.support_sets import SupportSets
    from warpedganspace_amd.trainer import TrainStep
    torch.manual_seed(0)          # identical random-init G / S / R on every rank; the sampler seed is derived per rank
    if gan == 'stylegan2':
        from warpedganspace_amd.gan_load import build_stylegan2
        G = build_stylegan2(None, resolution=size, shift_in_w_space=w_space)
    elif gan == 'proggan':
        from warpedganspace_amd.proggan import build_proggan
        G = build_proggan(None, num_blocks={1024: 18, 512: 16, 256: 14}[size])
    elif gan == 'biggan':
        from warpedganspace_amd.biggan import BigGANWrapper, Generator
        G = BigGANWrapper(Generator(G_ch=96, dim_z=120, shared_dim=128, hier=True, G_attn='64', BN_eps=1e-5, SN_eps=1e-6,
                                    resolution=size, n_classes=1000), (239,))
    else:
        raise ValueError(gan)
    S = SupportSets(K, N, G.dim_z, learn_alphas=False, learn_gammas=True, gamma=1.0 / G.dim_z)
    R = Reconstructor('ResNet', K, channels=3)
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    return TrainStep(G.to(dev).eval(), S.to(dev).train(), R.to(dev).train(), make_params(w_space), B, dev, world=world, seed=0,
                     rank=rank, precision=precision, r_precision=r_precision, **ENGINE_KW)


def timed_steps(eng, steps, warmup, world, dev):
    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        eng.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def conv_profile(eng, nprof=2):
    """HIP events around every implicit-GEMM launch of `nprof` extra steps.  torch's current stream IS the launch stream, and
    the steps run single-stream (with the un-shifted pass on the side stream, kernels of the other stream would run inside
    the event pairs and inflate the per-launch durations).  Returns a list of (shape label, kernel symbol, FLOPs, ms, launches)
    per step, one entry per (label, symbol)."""
    from warpedganspace_amd import _lib as L
    from warpedganspace_amd import conv as C
    torch.cuda.synchronize()
    eng._pre = None            # a batch whose un-shifted pass was generated one step ahead would make the first profiled step one
    C.PROFILE = []             # generator pass short: the profiled steps draw their own batches and run every launch themselves
    L.lib().wgs_dev_trace_kernels(1)
    two, eng.two_streams = eng.two_streams, False
    try:
        for _ in range(nprof):
            eng.step()
        torch.cuda.synchronize()
    finally:
        eng.two_streams = two
        L.lib().wgs_dev_trace_kernels(0)
        recs, C.PROFILE = C.PROFILE, None
    by = {}
    for kind, fl, s, e, sym in recs:
        k = by.setdefault((kind, sym), [0.0, 0.0, 0.0])
        k[0] += fl / nprof; k[1] += s.elapsed_time(e) / nprof; k[2] += 1.0 / nprof
    return [(kind, sym, v[0], v[1], v[2]) for (kind, sym), v in by.items()]


def _label_precision(label):
    parts = label.split()
    return parts[1] if len(parts) > 1 and parts[1] in MFMA_PER_PRODUCT else 'fp32'


def _executed_per_product(prec):
    """MFMA FLOPs a launch executes per algorithmic (direct-form) FLOP, capped at 1: only the Winograd form executes fewer
    multiplies than the direct form; the multi-MFMA schemes (bf16x3, f16x2) are rated on their algorithmic FLOPs."""
    return min(1.0, MFMA_PER_PRODUCT[prec])


def pmc_traffic(symbol, shapes, pmc_files=None):
    """Launch-weighted mean HBM bytes per launch of `symbol` over its launch shapes, measured and algorithmic, from the newest
    committed rocprofv3 --pmc summary (profiles/r*_conv_pmc.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes) that
    holds the symbol.  shapes: [{shape, launches}] of this run (weights; shapes without a PMC row are left out and named)."""
    import glob
    files = pmc_files if pmc_files is not None else sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_conv_pmc.json')), reverse=True)
    for f in files:
        try:
            rows = [r for r in json.load(open(f))['kernels'] if r.get('symbol') == symbol]
        except Exception:  # noqa: BLE001
            continue
        if not rows:
            continue
        by_shape = {r['shape']: r for r in rows}
        n = {sh['shape']: sh['launches'] for sh in shapes}
        used = [k for k in by_shape if k in n] or list(by_shape)
        wsum = sum(n.get(k, 1.0) for k in used)
        t = sum(n.get(k, 1.0) * by_shape[k]['hbm_bytes_per_launch'] for k in used) / wsum
        al = sum(n.get(k, 1.0) * by_shape[k]['algorithmic_bytes_per_launch'] for k in used) / wsum
        return {"traffic": round(t / 1e9, 3), "traffic_algorithmic": round(al / 1e9, 3), "source": os.path.relpath(f, REPO),
                "shapes_measured": len(used), "shapes_in_step": len(n)}
    return None


def roofline_of(recs, img_per_s_per_gpu, gflop_per_img, pmc_files=None):
    """Per-kernel-symbol accounting of the conv launches of a step.  The dominant kernel = the symbol with the largest summed
    time.  achieved = MFMA FLOPs it EXECUTES for its launches / its summed HIP-event time; frac = achieved / peak, <= 1 by
    construction; direct_equiv_TFLOPs = the algorithmic (direct-form) FLOPs over the same time."""
    sym = {}
    for label, s, fl, ms, n in recs:
        d = sym.setdefault(s or '?', {'fl': 0.0, 'ms': 0.0, 'n': 0.0, 'ex_fl': 0.0, 'mfma_fl': 0.0, 'prec': set(), 'shapes': []})
        prec = _label_precision(label)
        d['fl'] += fl; d['ms'] += ms; d['n'] += n
        d['ex_fl'] += _executed_per_product(prec) * fl          # what the roofline fraction is made of
        d['mfma_fl'] += MFMA_PER_PRODUCT[prec] * fl             # matrix-pipe occupancy (counts the 2 / 3 MFMAs of the split schemes)
        d['prec'].add(prec)
        d['shapes'].append({"shape": label, "gflop_per_step": round(fl / 1e9, 1), "ms_per_step": round(ms, 3),
                            "TFLOP/s": round(fl / ms / 1e9, 1), "launches": round(n, 1)})

    def peak_of(d):
        return FP32_MFMA_PEAK_TF if d['prec'] <= {'fp32', 'fp32w'} else F16_MFMA_PEAK_TF
    dom = max(sym, key=lambda k: sym[k]['ms'])
    d = sym[dom]
    peak = peak_of(d)
    tf = d['ex_fl'] / d['ms'] / 1e9
    a_fl, a_ex, a_ms = sum(v['fl'] for v in sym.values()), sum(v['ex_fl'] for v in sym.values()), sum(v['ms'] for v in sym.values())
    tr = pmc_traffic(dom, d['shapes'], pmc_files)
    by_symbol = [{"symbol": k, "ms_per_step": round(v['ms'], 3), "share_of_conv_time": round(v['ms'] / a_ms, 4), "launches": round(v['n'], 1),
                  "gflop_per_step": round(v['fl'] / 1e9, 1), "TFLOP/s": round(v['ex_fl'] / v['ms'] / 1e9, 1), "peak": peak_of(v),
                  "frac": round(v['ex_fl'] / v['ms'] / 1e9 / peak_of(v), 4),
                  "direct_equiv_TFLOPs": round(v['fl'] / v['ms'] / 1e9, 1)}
                 for k, v in sorted(sym.items(), key=lambda kv: -kv[1]['ms'])]
    out = {"bound": "mfma", "kernel": dom, "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
           "direct_equiv_TFLOPs": round(d['fl'] / d['ms'] / 1e9, 2),
           "traffic": tr["traffic"] if tr else None, "traffic_algorithmic": tr["traffic_algorithmic"] if tr else None,
           "traffic_source": ("static: " + tr["source"]) if tr else None,
           "traffic_note": ("GB per launch, launch-weighted mean over %d of the symbol's %d shapes; STATIC, from the committed rocprofv3 --pmc passes %s "
                            "(FETCH_SIZE x2 on gfx950 + WRITE_SIZE)" % (tr["shapes_measured"], tr["shapes_in_step"], tr["source"])) if tr else
                           "not measured for this symbol (PMC counters need separate rocprofv3 --pmc passes: profiles/)",
           "mfma_pipe_frac": round(d['mfma_fl'] / d['ms'] / 1e9 / peak, 4),
           "kernel_launches_per_step": round(d['n'], 1), "kernel_ms_per_step": round(d['ms'], 3),
           "kernel_avg_launch_ms": round(d['ms'] / d['n'], 4), "kernel_gflop_per_step": round(d['fl'] / 1e9, 1),
           "kernel_shapes": sorted(d['shapes'], key=lambda r: -r['ms_per_step']),
           "all_conv_launches": {"TFLOP/s": round(a_ex / a_ms / 1e9, 2), "direct_equiv_TFLOPs": round(a_fl / a_ms / 1e9, 2),
                                 "ms_per_step": round(a_ms, 3), "gflop_per_step": round(a_fl / 1e9, 1)},
           "by_symbol": by_symbol[:10],
           "method": "HIP events on the launch stream around every conv launch of 2 extra single-stream steps; per symbol: sum of the MFMA FLOPs "
                     "the launches execute (direct form: 2 * pixels * Cout * Cin * taps; Winograd F(2x2,3x3): 16/36 of that) / sum of their "
                     "durations; rocprofv3 tables of the same build: profiles/r5_step_*_kernel_stats.md"}
    if gflop_per_img:
        # the step's FLOPs: SURVEY.md section 8(d)'s algorithmic count, scaled by executed / algorithmic of the profiled conv launches
        alg = img_per_s_per_gpu * gflop_per_img / 1e3
        out["step_achieved_TFLOPs"] = round(alg * a_ex / a_fl, 2)
        out["step_frac"] = round(alg * a_ex / a_fl / peak, 4)
        out["step_direct_equiv_TFLOPs"] = round(alg, 2)
    return out


def host_overheads(eng, n=3):
    """Kernel launches per step issued by libwgs_hip.so (its own counter; torch's sampler / memset launches come on top: the
    rocprofv3 tables in profiles/ count everything) and the host time to ENQUEUE one step (queue drained before each)."""
    from warpedganspace_amd import _lib as L
    lib = L.lib()
    torch.cuda.synchronize()
    c0 = lib.wgs_dev_launch_count()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step()
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    c1 = lib.wgs_dev_launch_count()
    return {"library_launches_per_step": round((c1 - c0) / n, 1), "host_enqueue_ms_per_step": round(min(ts) * 1e3, 3),
            "host_enqueue_ms_per_step_mean": round(sum(ts) / n * 1e3, 3),
            "note": "enqueue = wall time of TrainStep.step() returning with the device queue empty at its start (Python + ctypes + HIP "
                    "launch calls of one step; the device runs behind it)"}


def hbm_subpaths(eng, dev, B):
    """Achieved GB/s of the HBM-bound sub-kernels of the step on their cfg3-sized operands:
    algorithmic bytes / HIP-event time of the standalone launch (on torch's current stream = the launch stream)."""
    from warpedganspace_amd import _lib as L
    lib, st = L.lib(), L.stream()

    def timed(fn, n=5, pre=None):
        fn(); torch.cuda.synchronize()
        tot = 0.0
        for _ in range(n):
            if pre is not None:
                pre()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            tot += a.elapsed_time(b)
        return tot / n * 1e-3

    out = {}
    S = eng.S
    K, n2 = S.ALPHAS.shape
    d = S.support_vectors_dim
    z = torch.randn(B, d, device=dev); idx = torch.randint(0, K, (B,), device=dev); mag = torch.rand(B, device=dev)
    shift = torch.empty(B, d, device=dev)
    lg = S.LOGGAMMA.reshape(-1)
    t = timed(lambda: L.check(lib.wgs_rbf_fwd(L.ptr(S.SUPPORT_SETS), L.ptr(S.ALPHAS), L.ptr(lg), L.c_float(float(S.gamma)),
                                              L.ptr(idx, torch.int64), L.ptr(z), L.ptr(mag), L.ptr(shift), L.ptr(eng.rbf_ws),
                                              B, K, n2, d, st), 'rbf'))
    by = B * ((n2 * d + n2 + 1 + d) * 4 + d * 4)
    out['rbf_fwd'] = {"bytes": by, "us": round(t * 1e6, 1), "GB/s": round(by / t / 1e9, 1), "note": "latency-bound: 4.3 MB per launch"}
    # RBF backward: re-reads the selected rows + z + gout, writes the selected rows' gradients (atomics; the dense zero-fill
    # of the [K, 2N*d] gradient is the bucket memset, counted with Adam's traffic below)
    gout = torch.randn(B, d, device=dev)
    dtable = torch.zeros_like(S.SUPPORT_SETS); dlg = torch.zeros(K, device=dev)
    t = timed(lambda: L.check(lib.wgs_rbf_bwd(L.ptr(S.SUPPORT_SETS), L.ptr(S.ALPHAS), L.ptr(lg), L.c_float(float(S.gamma)),
                                              L.ptr(idx, torch.int64), L.ptr(z), L.ptr(mag), L.ptr(gout), L.ptr(eng.rbf_ws),
                                              L.ptr(dtable), L.ptr(dlg), None, None, B, K, n2, d, st), 'rbf_bwd'))
    by = B * ((2 * n2 * d + n2 + 1 + 3 * d) * 4)
    out['rbf_bwd'] = {"bytes": by, "us": round(t * 1e6, 1), "GB/s": round(by / t / 1e9, 1), "note": "latency-bound; rows read once, gradient rows written once"}
    # Adam on the flat [R | S] bucket: 4 reads + 3 writes per element
    bk = eng.bucket
    n = bk.flat.numel()
    sc = bk.step_count
    t = timed(lambda: bk.adam_step(world=1))
    bk.step_count = sc
    out['adam'] = {"bytes": 7 * 4 * n, "us": round(t * 1e6, 1), "GB/s": round(7 * 4 * n / t / 1e9, 1)}
    # blur + noise + bias + lrelu of the 256x256 up-conv: [B,257,257,128] -> [B,256,256,128]
    C_, H = 128, 256
    tin = torch.randn(B, H + 1, H + 1, C_, device=dev); y = torch.empty(B, H, H, C_, device=dev)
    k4 = torch.ones(4, 4, device=dev) / 16; nz = torch.randn(H * H, device=dev); nw = torch.ones(1, device=dev); bias = torch.zeros(C_, device=dev)
    t = timed(lambda: L.check(lib.wgs_sg2_blur_noise_bias_act(L.ptr(tin), L.ptr(k4), L.ptr(nz), L.ptr(nw), L.ptr(bias), L.ptr(y), None,
                                                              B, H, H, C_, st), 'blur'))
    by = (tin.numel() + y.numel()) * 4
    out['blur_noise_bias_act_256'] = {"bytes": by, "us": round(t * 1e6, 1), "GB/s": round(by / t / 1e9, 1)}
    # ToRGB at 256x256: reads [B,65536,128], writes [B,3,65536]
    s_ = torch.randn(B, C_, device=dev); w_ = torch.randn(3, C_, device=dev); b3 = torch.zeros(3, device=dev)
    img = torch.empty(B, 3, H * H, device=dev)
    t = timed(lambda: L.check(lib.wgs_sg2_torgb_fwd(L.ptr(y), L.ptr(s_), L.ptr(w_), L.ptr(b3), None, L.ptr(img), B, H * H, C_,
                                                    L.c_float(1.0), st), 'torgb'))
    by = (y.numel() + img.numel()) * 4
    out['torgb_256'] = {"bytes": by, "us": round(t * 1e6, 1), "GB/s": round(by / t / 1e9, 1)}
    # train-mode BatchNorm forward (+ReLU) of ResNet conv1's output [B*128*128, 64]: stats pass + apply pass = 2 reads + 1 write
    N_, Cb = B * 128 * 128, 64
    xb = torch.randn(N_, Cb, device=dev); yb = torch.empty_like(xb)
    g_, b_ = torch.ones(Cb, device=dev), torch.zeros(Cb, device=dev)
    mean, invstd = torch.empty(Cb, device=dev), torch.empty(Cb, device=dev)
    ws = torch.zeros(64 * Cb, dtype=torch.float64, device=dev)
    t = timed(lambda: L.check(lib.wgs_bn_fwd(L.ptr(xb), L.ptr(g_), L.ptr(b_), None, L.ptr(yb), L.ptr(mean), L.ptr(invstd), None, None,
                                             None, L.rawptr(ws), L.c_int64(N_), Cb, L.c_float(1e-5), L.c_float(0.1), 1, 1, st), 'bn'))
    by = 3 * xb.numel() * 4
    out['bn_fwd_relu_conv1'] = {"bytes": by, "us": round(t * 1e6, 1), "GB/s": round(by / t / 1e9, 1)}
    out['peak'] = {"GB/s": 8000.0, "achievable_GB/s": 6300.0, "source": "MI355X_MICROARCH.md"}
    return out


def physical_cores():
    """Physical cores this process may run on: distinct (package, core) pairs of the allowed logical CPUs (sysfs topology);
    falls back to the allowed logical CPUs."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    cores = set()
    for c in allowed:
        try:
            base = '/sys/devices/system/cpu/cpu%d/topology/' % c
            cores.add((open(base + 'physical_package_id').read().strip(), open(base + 'core_id').read().strip()))
        except OSError:
            return len(allowed)
    return max(1, len(cores))


def cpu_baseline(size, K, N, b, steps, threads):
    """The reference step AS WRITTEN (incl. the generator's unused weight gradients) replayed by the oracle
    with plain PyTorch-CPU ops on this box's host cores.  Bounded sample."""
    from oracle import wgs_oracle as O
    from warpedganspace_amd.reconstructor import Reconstructor
    from warpedganspace_amd.stylegan2 import Generator
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    # tiny warm-up (thread pool, allocator) on a 32x32 generator so that the timed sample stays bounded
    w = O.ReferenceStep({k: v.detach().clone() for k, v in Generator(32, 512, 8).state_dict().items()},
                        O.support_sets_init(K, N, 512, 1.0 / 512),
                        {k: v.detach().clone().contiguous() for k, v in Reconstructor('ResNet', K).state_dict().items()},
                        32, learn_gammas=True, gamma=1.0 / 512, g_requires_grad=True)
    w.step(torch.randn(2, 512), torch.randint(0, K, (2,)), torch.rand(2) * 0.2 + 0.25)
    sd_g = {k: v.detach().clone() for k, v in Generator(size, 512, 8).state_dict().items()}
    sd_s = O.support_sets_init(K, N, 512, 1.0 / 512)
    sd_r = {k: v.detach().clone().contiguous() for k, v in Reconstructor('ResNet', K).state_dict().items()}
    ref = O.ReferenceStep(sd_g, sd_s, sd_r, size, learn_gammas=True, gamma=1.0 / 512, g_requires_grad=True)
    # all physical cores first; on a 2 x 64-core host PyTorch-CPU's convolutions at batch 4 run SLOWER on 128 threads than
    # on 32 (27.5 s against 10.6 s per step, round 5), so the same step is also timed on min(32, cores) threads and the better rate is the
    # baseline — both are stated
    counts = [threads] + ([32] if threads > 32 else [])
    g = torch.Generator().manual_seed(1)
    z = torch.randn(b, 512, generator=g)
    idx = torch.randint(0, K, (b,), generator=g)
    mag = (torch.rand(b, generator=g) * 0.2 + 0.25)
    by = {}
    for n in counts:
        torch.set_num_threads(n)
        ts = []
        for it in range(steps):
            t0 = time.time()
            ref.step(z, idx, mag)
            ts.append(time.time() - t0)
        by[n] = ts
    best = min(by, key=lambda n: sum(by[n]) / len(by[n]))
    # ... and at least two steps on the thread count that is reported (one step per count has no spread)
    torch.set_num_threads(best)
    while len(by[best]) < max(2, steps):
        t0 = time.time()
        ref.step(z, idx, mag)
        by[best].append(time.time() - t0)
    dt = sum(by[best]) / len(by[best])
    return {"value": round(b / dt, 4), "unit": "images/sec", "cores": best, "kind": "port", "physical_cores": threads,
            "sample": "%d step(s) of batch %d per thread count, >= 2 on the reported one (after a 32x32 warm-up step), StyleGAN2-%d K=%d N=%d ResNet-18, reference step as written "
                      "(lib/trainer.py:190-254) replayed by oracle/wgs_oracle.py on PyTorch-CPU; per-step seconds by threads: %s"
                      % (steps, b, size, K, N, {n: [round(t, 2) for t in v] for n, v in by.items()})}


# Short runs of the other arithmetic modes and BASELINE configs (N = 1 only).
# (name, gan, size, K, N, batch, generator precision, reconstructor precision, w_space, steps, GFLOP-per-image key or None, short key of the line's `others_images_per_sec`)
EXTRA = [
    ("cfg3 StyleGAN2-256 bf16x3", 'stylegan2', 256, 128, 32, 32, 'bf16x3', 'auto', False, 10, 'stylegan2-256', 'cfg3_bf16x3'),
    ("cfg3 StyleGAN2-256 f16", 'stylegan2', 256, 128, 32, 32, 'f16', 'auto', False, 10, 'stylegan2-256', 'cfg3_f16'),
    ("cfg3 StyleGAN2-256 f16x2", 'stylegan2', 256, 128, 32, 32, 'f16x2', 'auto', False, 10, 'stylegan2-256', 'cfg3_f16x2'),
    ("cfg3 StyleGAN2-256 bf16x3w (split-bf16, F(2,3) form of the stride-1 3x3 convs)", 'stylegan2', 256, 128, 32, 32, 'bf16x3w', 'auto', False, 10, 'stylegan2-256', 'cfg3_bf16x3w'),
    ("cfg3 StyleGAN2-256 mixed: the UN-calibrated default table (reported only: on this initialisation single images sit on the 1e-3 gate)", 'stylegan2', 256, 128, 32, 32, 'mixed', 'auto', False, 30, 'stylegan2-256', 'cfg3_mixed_uncalibrated'),
    ("cfg3 StyleGAN2-256 auto, reconstructor convs in exact fp32 [R fp32]", 'stylegan2', 256, 128, 32, 32, 'auto', 'fp32', False, 20, 'stylegan2-256', 'cfg3_auto_Rfp32'),
    ("cfg3 StyleGAN2-256 auto, W-space", 'stylegan2', 256, 128, 32, 32, 'auto', 'auto', True, 10, 'stylegan2-256', 'cfg3_auto_Wspace'),
    ("cfg2 ProgGAN native 1024, K=64 N=16 B=32, auto", 'proggan', 1024, 64, 16, 32, 'auto', 'auto', False, 12, 'proggan-1024', 'cfg2_proggan1024_auto'),
    ("cfg2' ProgGAN truncated to 256 (first 14 blocks), K=64 N=16 B=32, auto", 'proggan', 256, 64, 16, 32, 'auto', 'auto', False, 8, 'proggan-256', 'cfg2_proggan256_auto'),
    ("cfg4 BigGAN-128 (the reference's architecture), K=128 N=32 B=16, auto", 'biggan', 128, 128, 32, 16, 'auto', 'auto', False, 8, 'biggan-128', 'cfg4_biggan128_auto'),
    ("cfg4' BigGAN-256 (generator_arch 256, class-conditional), K=128 N=32 B=16, auto", 'biggan', 256, 128, 32, 16, 'auto', 'auto', False, 6, None, 'cfg4_biggan256_auto'),
    ("cfg5 StyleGAN2-1024, K=200 N=64 B=8, fp32 with the Winograd form (fp32w)", 'stylegan2', 1024, 200, 64, 8, 'fp32w', 'auto', False, 6, 'stylegan2-1024', 'cfg5_sg1024_fp32w'),
    ("cfg5 StyleGAN2-1024, K=200 N=64 B=8, direct-form exact fp32", 'stylegan2', 1024, 200, 64, 8, 'fp32', 'auto', False, 3, 'stylegan2-1024', 'cfg5_sg1024_fp32'),
    ("cfg5 StyleGAN2-1024, K=200 N=64 B=8, 16-bit MFMA path = auto (the table the engine calibrates on its generator; on this initialisation: split-bf16, the stride-1 layers in the F(2,3) form)", 'stylegan2', 1024, 200, 64, 8, 'auto', 'auto', False, 12, 'stylegan2-1024', 'cfg5_sg1024_auto'),
    ("cfg5 StyleGAN2-1024, K=200 N=64 B=8, bf16x3 everywhere", 'stylegan2', 1024, 200, 64, 8, 'bf16x3', 'auto', False, 6, 'stylegan2-1024', 'cfg5_sg1024_bf16x3'),
]


PRECISION_CHECK_IMAGES = 2304       # DESIGN.md section 3.2's sample size (there: 4 weight fills x 576 codes; here: THIS engine's weights)


def precision_check(eng, images=PRECISION_CHECK_IMAGES):
    """TrainStep.check_precision() on the timed engine: the measured image error of the arithmetic the run was timed in,
    against the exact-fp32 kernels on the same weights — max-norm relative, per batch tensor and per single image."""
    r = eng.check_precision(batches=max(1, min(72, images // eng.B)))
    if r is None:
        return None
    return {"batch_max": float('%.3g' % r['batch']), "image_median": float('%.3g' % r['per_image_median']), "image_p99": float('%.3g' % r['per_image_p99']),
            "image_max": float('%.3g' % r['per_image_max']), "over_gate_frac": round(r['over_gate_frac'], 5), "gate": r['gate'], "n": r['n']}


def run_one(dev, gan, size, K, N, B, prec, r_prec, w_space, steps, warmup, gkey, world=1, rank=0, full=False, check=False, eng_world=None):
    """Build an engine, time `steps` steps after `warmup`, profile the conv launches; `full` adds the roofline and host sections;
    `check` the measured image error of the engine's arithmetic; eng_world=1 inside an N > 1 job: a single-rank engine (no collectives)."""
    from warpedganspace_amd import conv as C
    eng = build(dev, gan, K, N, B, rank=rank, w_space=w_space, size=size, precision=prec, r_precision=r_prec, world=eng_world)
    name = C.precision_name(eng.precision)
    if eng_world is not None:
        world = eng_world
    dt = timed_steps(eng, steps, warmup, world, dev)
    recs = conv_profile(eng, 2 if full else 1)
    a_fl, a_ms = sum(r[2] for r in recs), sum(r[3] for r in recs)
    rec = {"precision": name, "r_arith": list(eng.r_arith), "dtype": DTYPE[name], "value": round(B * world * steps / dt, 2), "unit": "images/sec",
           "n_gpus": world, "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "warmup": warmup,
           "conv_TFLOP/s": round(a_fl / a_ms / 1e9, 1), "conv_ms_per_step": round(a_ms, 2), "conv_gflop_per_step": round(a_fl / 1e9, 1)}
    if gkey and gkey in GFLOP_PER_IMG:
        rec["step_achieved_TFLOPs"] = round(B * steps / dt * GFLOP_PER_IMG[gkey] / 1e3, 1)
    if full:
        rec["dtype_detail"] = DTYPE_TEXT[name] + R_TEXT.get(tuple(eng.r_arith), "; reconstructor arithmetic (forward, dgrad, wgrad; 0 exact fp32, 1 split-bf16 x3): %s" % (tuple(eng.r_arith),))
        rec["roofline"] = roofline_of(recs, B * steps / dt, GFLOP_PER_IMG.get(gkey))
        rec["host"] = host_overheads(eng)
    if getattr(eng, 'strict_calibration', None):
        rec["strict_calibration"] = eng.strict_calibration
    if check:
        try:
            rec["precision_check"] = precision_check(eng)
        except Exception as e:  # noqa: BLE001
            rec["precision_check"] = {"error": repr(e)[:200]}
    return rec, eng


def run_extra(dev):
    out = []
    for name, gan, size, K, N, B, prec, r_prec, w_space, steps, gkey, short in EXTRA:
        try:
            rec, eng = run_one(dev, gan, size, K, N, B, prec, r_prec, w_space, steps, 6, gkey, check=(prec == 'mixed'))
            out.append(dict({"config": name, "key": short}, **rec))
            del eng
        except Exception as e:  # noqa: BLE001
            out.append({"config": name, "key": short, "precision": prec, "error": repr(e)[:300]})
        torch.cuda.empty_cache()
    return out


def spawn_ranks(n, argv, backend):
    """--gpus N without a launcher: start N ranks of this script (one per GPU) and relay their output."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and backend != 'gloo':
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node" % (n, have))
    if have < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))


def comm_section(eng, backend, value=None):
    eng.comm_events = []
    for _ in range(5):
        eng.step()
    torch.cuda.synchronize()
    waits = [a.elapsed_time(b) for a, b in eng.comm_events]
    eng.comm_events = None
    return {"backend": "RCCL (torch.distributed 'nccl')" if backend == 'nccl' else "gloo (development switch: ranks may share one device)",
            "world_size_observed": dist.get_world_size(), "allreduce_bytes_per_step": eng.allreduce_bytes, "collectives_per_step": 2,
            "exposed_wait_ms_per_step": round(sum(waits) / len(waits), 3),
            "per_gpu_images_per_sec": round(value / dist.get_world_size(), 2) if value else None,
            "note": "main-stream time between reaching the wait for the two all-reduces (R group, queued behind R's weight gradients on "
                    "the side stream; S group, after the RBF backward) and their completion, mean of 5 extra steps on rank 0"}


def n1_reference(dev, args, prec, r_prec, gkey, rank, world):
    """Inside an N > 1 job: the same workload on ONE rank with a single-rank engine while the other ranks idle at a barrier
    (VERDICT r4 #6c: the N = 1 rate of the same process, box and build beside the N > 1 one)."""
    out = None
    dist.barrier()
    if rank == 0:
        try:
            rec, eng = run_one(dev, args.gan, args.size, args.K, args.N, args.batch, prec, r_prec, args.w_space,
                               min(args.steps, 30), min(args.warmup, 10), gkey, rank=0, eng_world=1)
            out = {"value": rec["value"], "ms_per_step": rec["ms_per_step"], "steps": rec["steps"]}
            del eng
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            out = {"error": repr(e)[:200]}
    dist.barrier()
    return out


def full_record(args, world, head, stats, comm, hbm, extra, cpu, gkey):
    """Everything this run measured (the side file); final_line() condenses it to the < 4 KB line the driver parses."""
    arch = {'stylegan2': 'StyleGAN2-FFHQ-%d' % args.size, 'proggan': 'ProgGAN (%d)' % args.size, 'biggan': 'BigGAN-%d' % args.size}[args.gan]
    return {"metric": "training images/sec (warp->G->R->loss) %s K=%d" % ({'stylegan2': 'StyleGAN2-%d' % args.size}.get(args.gan, arch), args.K),
            "value": head["value"], "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": head["dtype"], "dtype_detail": head.get("dtype_detail"),
            "data": "synthetic (random-init weights, z ~ N(0,I) sampled on the device)",
            "config": {"workload": "%s arch, K=%d, N=%d, ResNet-18 R, batch %d/GPU, %s-space, learn_gammas"
                                   % (arch, args.K, args.N, args.batch, 'W' if args.w_space else 'Z'),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "precision": head["precision"],
                       "precision_requested": args.precision, "r_arith": head["r_arith"],
                       "algorithmic_gflop_per_image": GFLOP_PER_IMG.get(gkey)},
            "last_stats": stats, "roofline": head.get("roofline"), "host": head.get("host"), "comm": comm, "hbm_subpaths": hbm,
            "extra": extra or None, "cpu_baseline": cpu}


ROOFLINE_LINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_algorithmic", "traffic_source", "direct_equiv_TFLOPs",
                      "kernel_launches_per_step", "kernel_avg_launch_ms", "step_achieved_TFLOPs", "step_frac", "step_direct_equiv_TFLOPs")
LINE_LIMIT = 4096


def _short_run(e):
    """One same-workload companion run (product default / direct fp32) as the line carries it."""
    r = e.get("roofline") or {}
    out = {"precision": e.get("precision"), "value": e.get("value"), "ms_per_step": e.get("ms_per_step"), "dtype": e.get("dtype")}
    if r:
        out["roofline"] = {k: r.get(k) for k in ("kernel", "achieved", "peak", "frac", "step_frac")}
    c = e.get("comm")
    if c:
        out["exposed_wait_ms_per_step"] = c.get("exposed_wait_ms_per_step")
        out["per_gpu_images_per_sec"] = c.get("per_gpu_images_per_sec")
        out["n1_same_job"] = c.get("n1_same_job")
    if e.get("precision_check"):
        out["precision_check"] = e["precision_check"]
    sc = e.get("strict_calibration")
    if sc:       # the per-layer table the engine calibrated on its own generator ('auto' = 'mixed-strict' for StyleGAN2)
        out["table"] = sc.get("table")
        out["fp16_layers"] = sc.get("fp16_layers")
    return out


def final_line(full, extra_file):
    """The ONE stdout line the driver parses: < LINE_LIMIT bytes whatever the run measured (tests/test_bench_tables_cpu.py)."""
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    line["config"] = full["config"]
    r = full.get("roofline")
    line["roofline"] = {k: r.get(k) for k in ROOFLINE_LINE_KEYS if k in r} if r else None
    c = full.get("cpu_baseline")
    if c:
        c = dict(c)
        if isinstance(c.get("sample"), str) and len(c["sample"]) > 300:
            c["sample"] = c["sample"][:297] + "..."
        if "error" in c:
            c["error"] = str(c["error"])[:200]
    line["cpu_baseline"] = c
    cm = full.get("comm")
    line["comm"] = {k: cm.get(k) for k in ("backend", "world_size_observed", "allreduce_bytes_per_step", "collectives_per_step",
                                           "exposed_wait_ms_per_step", "per_gpu_images_per_sec", "n1_same_job")} if cm else None
    h = full.get("host")
    if h:
        line["host"] = {k: h.get(k) for k in ("library_launches_per_step", "host_enqueue_ms_per_step")}
    extra = full.get("extra") or []
    same = [e for e in extra if str(e.get("config", "")).startswith("headline workload")]
    for e in same:
        line["direct_fp32" if "direct-form" in e["config"] else "product"] = _short_run(e)
    others = {}
    for e in extra:
        if e in same:
            continue
        if e.get("key") == "cfg3_mixed_uncalibrated" and "product" in line and "error" not in e:
            pc = e.get("precision_check") or {}
            line["product"]["uncalibrated"] = {"precision": e.get("precision"), "value": e.get("value"), "ms_per_step": e.get("ms_per_step"),
                                               "precision_check": {k: pc.get(k) for k in ("batch_max", "image_p99", "image_max", "over_gate_frac", "n")} if pc else None}
            continue
        others[str(e.get("key") or e.get("config", "?"))[:40]] = e.get("value") if "error" not in e else "error"
    line["others_images_per_sec"] = others or None
    line["extra_file"] = extra_file
    s = json.dumps(line)
    # belt and braces: shed the optional sections, largest first, until the line fits
    for k in ("others_images_per_sec", "direct_fp32", "host", "product"):
        if len(s) < LINE_LIMIT:
            break
        line.pop(k, None)
        s = json.dumps(line)
    if len(s) >= LINE_LIMIT:
        raise RuntimeError("bench.py: final line is %d bytes (limit %d)" % (len(s), LINE_LIMIT))
    return s


def main():
    from warpedganspace_amd import conv as C
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch')
    ap.add_argument('--gan', choices=('stylegan2', 'proggan', 'biggan'), default='stylegan2')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('-K', type=int, default=128)
    ap.add_argument('-N', type=int, default=32)
    ap.add_argument('--w-space', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=4)
    ap.add_argument('--cpu-steps', type=int, default=1)
    ap.add_argument('--cpu-threads', type=int, default=0, help="threads of the CPU baseline (0 = every physical core this process may run on)")
    ap.add_argument('--precision', choices=tuple(C.PRECISION_NAMES), default='fp32w',
                    help="arithmetic of the HEADLINE run's generator convs (default: fp32w = fp32, Winograd form of the 3x3 stride-1 convs)")
    ap.add_argument('--no-direct-run', action='store_true', help="skip extra[1] (the direct-form exact-fp32 run with the same steps / warmup)")
    ap.add_argument('--r-precision', choices=['fp32', 'fp32w', 'bf16x3', 'auto'], default='auto',
                    help="arithmetic of the Reconstructor's convs (auto = exact fp32 beside an fp32 generator, split-bf16 x3 beside a 16-bit one)")
    ap.add_argument('--product-precision', choices=tuple(C.PRECISION_NAMES), default=C.DEFAULT_PRECISION,
                    help="arithmetic of the extra[0] run (same workload, same steps / warmup): the product's default")
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--single-stream', action='store_true', help='no side streams (profiling: kernel times then add up to the step)')
    ap.add_argument('--no-product-run', action='store_true', help='skip extra[0] (the default-arithmetic run with the same steps / warmup)')
    ap.add_argument('--no-n1-reference', action='store_true', help="N > 1: skip the single-rank reference runs on rank 0 (comm.n1_same_job)")
    ap.add_argument('--no-extra', action='store_true', help='skip the short runs of the other arithmetic modes / configs')
    ap.add_argument('--extra-out', default=os.path.join('gpurun_out', 'bench_extra.json'),
                    help="side file for everything the < 4 KB stdout line leaves out (relative to the repo root)")
    ap.add_argument('--dist-backend', choices=('nccl', 'gloo'), default=os.environ.get('WGS_DIST_BACKEND', 'nccl'),
                    help="nccl = RCCL (one rank per GPU); gloo = development switch, ranks share the visible device(s)")
    args = ap.parse_args()
    if args.single_stream:
        ENGINE_KW['two_streams'] = False

    if os.environ.get('WGS_STEM') == '0':            # development A/B: the Reconstructor's stem in the gather form
        from warpedganspace_amd import reconstructor as _RR
        _RR.STEM_S2D = False
    if os.environ.get('WGS_SPLIT') == '0':           # development A/B: the prefetched pass in one piece behind the shifted forward
        from warpedganspace_amd import trainer as _T
        _T.TrainStep.split_prefetch_default = False
    if os.environ.get('WGS_EAGER') == '1':           # development A/B: R's weight gradients as soon as their dy exists, next to R's own backward
        from warpedganspace_amd import trainer as _T3
        _T3.TrainStep.eager_wgrad_default = True
    if os.environ.get('WGS_PAUSE_RES'):
        from warpedganspace_amd import trainer as _T2
        _T2.TrainStep.split_pause_res_default = int(os.environ['WGS_PAUSE_RES'])
    if os.environ.get('WGS_TAIL'):                   # development A/B: tail stage of the prefetched pass: '0' off, 'pause,hook' resolutions
        from warpedganspace_amd import trainer as _T4
        if os.environ['WGS_TAIL'] == '0':
            _T4.TrainStep.tail_prefetch_default = False
        else:
            _pr, _hr = os.environ['WGS_TAIL'].split(',')
            _T4.TrainStep.tail_pause_res_default, _T4.TrainStep.tail_hook_res_default = int(_pr), int(_hr)
    if os.environ.get('WGS_RGB') == '0':             # development A/B: ToRGB as its own launch everywhere
        C.RGB_FUSED = False
    if os.environ.get('WGS_PRIO') == '0':            # development A/B: no high-priority stream for the step's critical path
        ENGINE_KW['priority_main'] = False
    if os.environ.get('WGS_FWD_PLANE') == '0':       # development A/B: forward fp16 planes off (conv.FWD_PLANE)
        C.FWD_PLANE = False
        C.DY_PLANE_MIN_CO = 256
    in_job = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    if args.gpus > 1 and not in_job:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], args.dist_backend))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.dist_backend == 'gloo':
        local_rank %= ndev
    if ndev <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, ndev))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    pin = None
    if world > 1 and args.dist_backend == 'nccl':
        # one launch thread per rank: keep it on the CPUs of its GPU's NUMA node, apart from the other ranks' 
        from warpedganspace_amd.hostpin import pin_rank
        pin = pin_rank(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)   # "nccl" is RCCL on ROCm

    gkey = '%s-%d' % (args.gan, args.size)
    full = not args.no_roofline
    head, eng = run_one(dev, args.gan, args.size, args.K, args.N, args.batch, args.precision, args.r_precision, args.w_space,
                        args.steps, args.warmup, gkey, world=world, rank=rank, full=full)
    stats = eng.pop_stats()
    comm = comm_section(eng, args.dist_backend, head["value"]) if world > 1 else None
    hbm = None
    if rank == 0 and world == 1 and full:
        try:
            hbm = hbm_subpaths(eng, dev, args.batch)
        except Exception as e:  # noqa: BLE001
            hbm = {"error": repr(e)}
    del eng
    torch.cuda.empty_cache()
    if world > 1 and not args.no_n1_reference:
        comm["n1_same_job"] = n1_reference(dev, args, args.precision, args.r_precision, gkey, rank, world)

    extra = []
    if not args.no_product_run:
        # the same workload, steps and warm-up in the product's default arithmetic, with its own roofline (every rank takes part)
        prod, eng = run_one(dev, args.gan, args.size, args.K, args.N, args.batch, args.product_precision, 'auto', args.w_space,
                            args.steps, args.warmup, gkey, world=world, rank=rank, full=full, check=True)
        prod["last_stats"] = eng.pop_stats()
        if world > 1:
            prod["comm"] = comm_section(eng, args.dist_backend, prod["value"])
        del eng
        torch.cuda.empty_cache()
        if world > 1 and not args.no_n1_reference:
            prod["comm"]["n1_same_job"] = n1_reference(dev, args, args.product_precision, 'auto', gkey, rank, world)
        extra.append(dict({"config": "headline workload in the product's default arithmetic (--precision %s), same steps / warmup"
                                     % args.product_precision}, **prod))
    if not args.no_direct_run and not args.no_product_run and args.precision == 'fp32w':
        dirr, eng = run_one(dev, args.gan, args.size, args.K, args.N, args.batch, 'fp32', 'fp32', args.w_space,
                            args.steps, args.warmup, gkey, world=world, rank=rank, full=full)
        dirr["last_stats"] = eng.pop_stats()
        extra.append(dict({"config": "headline workload in direct-form exact fp32 (--precision fp32: no Winograd), same steps / warmup"}, **dirr))
        del eng
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_extra:
        extra += run_extra(dev)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.gan == 'stylegan2':
        try:
            cpu = cpu_baseline(args.size, args.K, args.N, args.cpu_batch, args.cpu_steps, args.cpu_threads or physical_cores())
        except Exception as e:  # noqa: BLE001
            cpu = {"error": repr(e)}

    if rank == 0:
        doc = full_record(args, world, head, stats, comm, hbm, extra, cpu, gkey)
        doc["host_pinning_rank0"] = pin
        if world > 1:
            from warpedganspace_amd.hostpin import plan_all
            doc["host_pinning_plan"] = plan_all(int(os.environ.get('LOCAL_WORLD_SIZE', world)))
        path = args.extra_out if os.path.isabs(args.extra_out) else os.path.join(REPO, args.extra_out)
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, 'w') as f:
                json.dump(doc, f, indent=1)
        except OSError as e:
            path = "unwritable: %r" % (e,)
        print(final_line(doc, os.path.relpath(path, REPO) if os.path.isabs(path) else path), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
