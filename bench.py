#!/usr/bin/env python
"""bench.py — training-step images/sec (warp -> G -> R -> loss -> backward -> Adam) on MI355X.

Workload (BASELINE.json `metric`, configs[2]): StyleGAN2-FFHQ-256 architecture (random-init weights drawn
exactly as the reference constructors do; no checkpoints offline), K=128 warping functions x N=32 dipoles,
ResNet-18 reconstructor, batch 32 per GPU, Z-space shifts, --learn-gammas, synthetic z ~ N(0, I).

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one rank per GPU (torch.distributed.run); gradients of R and S are
all-reduced over RCCL once per step; per-GPU batch is fixed (weak scaling).

Prints ONE JSON line on rank 0: metric/value (whole-job images/sec), ms_per_step, `roofline` for the dominant
kernel family (implicit-GEMM MFMA conv: algorithmic FLOPs of its launches / their HIP-event durations,
against the fp32-MFMA peak) and `cpu_baseline` (the oracle's replay of the reference step, as written,
on this box's host cores — a bounded sample).
"""
import argparse
import json
import os
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

GFLOP_PER_IMG = 285.8          # G fwd x2 + G dgrad + R fwd + R bwd, StyleGAN2-256 / ResNet-18
FP32_MFMA_PEAK_TF = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)


def build(dev, size, K, N, B, seed, w_space=False, rank=0):
    from warpedganspace_amd.gan_load import build_stylegan2
    from warpedganspace_amd.reconstructor import Reconstructor
    from warpedganspace_amd.support_sets import SupportSets
    from warpedganspace_amd.trainer import TrainStep
    torch.manual_seed(0)          # identical random-init G / S / R on every rank; `seed` only drives the per-rank sampling
    G = build_stylegan2(None, resolution=size, shift_in_w_space=w_space)
    S = SupportSets(K, N, G.dim_z, learn_alphas=False, learn_gammas=True, gamma=1.0 / G.dim_z)
    R = Reconstructor('ResNet', K, channels=3)
    params = types.SimpleNamespace(reconstructor_lr=1e-4, support_set_lr=1e-4, min_shift_magnitude=0.25,
                                   max_shift_magnitude=0.45, lambda_cls=1.0, lambda_reg=0.25, z_truncation=None,
                                   shift_in_w_space=w_space)
    world = dist.get_world_size() if dist.is_initialized() else 1
    eng = TrainStep(G.to(dev).eval(), S.to(dev).train(), R.to(dev).train(), params, B, dev, world=world, seed=seed, rank=rank)
    return eng


def hbm_subpaths(eng, dev, B):
    """Achieved GB/s of the HBM-bound sub-kernels of the step on their cfg3-sized operands:
    algorithmic bytes / HIP-event time of the standalone launch (on torch's current stream = the launch stream)."""
    import ctypes
    from warpedganspace_amd import _lib as L
    lib, st = L.lib(), L.stream()

    def timed(fn, n=5):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e-3

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
    t = timed(lambda: L.check(lib.wgs_sg2_blur_noise_bias_act(L.ptr(tin), L.ptr(k4), L.ptr(nz), L.ptr(nw), L.ptr(bias), L.ptr(y),
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
    ws = torch.empty(64 * Cb, dtype=torch.float64, device=dev)
    t = timed(lambda: L.check(lib.wgs_bn_fwd(L.ptr(xb), L.ptr(g_), L.ptr(b_), None, L.ptr(yb), L.ptr(mean), L.ptr(invstd), None, None,
                                             None, L.rawptr(ws), L.c_int64(N_), Cb, L.c_float(1e-5), L.c_float(0.1), 1, 1, st), 'bn'))
    by = 3 * xb.numel() * 4
    out['bn_fwd_relu_conv1'] = {"bytes": by, "us": round(t * 1e6, 1), "GB/s": round(by / t / 1e9, 1)}
    out['peak'] = {"GB/s": 8000.0, "achievable_GB/s": 6300.0, "source": "MI355X_MICROARCH.md"}
    return out


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
    g = torch.Generator().manual_seed(1)
    times = []
    for it in range(steps):
        z = torch.randn(b, 512, generator=g)
        idx = torch.randint(0, K, (b,), generator=g)
        mag = (torch.rand(b, generator=g) * 0.2 + 0.25)
        t0 = time.time()
        ref.step(z, idx, mag)
        times.append(time.time() - t0)
    dt = sum(times) / len(times)
    return {"value": round(b / dt, 4), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "%d step(s) of batch %d (after a 32x32 warm-up step), StyleGAN2-%d K=%d N=%d ResNet-18, reference step as written "
                      "(lib/trainer.py:190-254) replayed by oracle/wgs_oracle.py on PyTorch-CPU" % (steps, b, size, K, N)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('-K', type=int, default=128)
    ap.add_argument('-N', type=int, default=32)
    ap.add_argument('--w-space', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=4)
    ap.add_argument('--cpu-steps', type=int, default=2)
    ap.add_argument('--cpu-threads', type=int, default=32)
    ap.add_argument('--r-precision', choices=['fp32', 'bf16x3'], default='fp32', help='arithmetic of the Reconstructor convs (default exact fp32)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--precision', choices=('bf16x3', 'fp32', 'f16', 'f16x2'), default='bf16x3',
                    help="arithmetic of the implicit-GEMM convs: split-bf16 x3 MFMA (fp32-class, ~1e-5) or exact fp32 MFMA")
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)   # "nccl" is RCCL on ROCm

    from warpedganspace_amd import conv as C
    C.set_precision(args.precision)
    from warpedganspace_amd import reconstructor as RR
    RR.R_PRECISION = 1 if args.r_precision == 'bf16x3' else 0
    eng = build(dev, args.size, args.K, args.N, args.batch, seed=0, rank=rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stats = eng.pop_stats()
    ms_per_step = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt

    roofline = None
    if not args.no_roofline:
        # HIP events around every implicit-GEMM launch (torch's current stream IS the launch stream)
        # (single stream for these two steps: with the un-shifted pass on the side stream, kernels of the other stream would
        # run inside the event pairs and inflate the per-launch durations)
        C.PROFILE = []
        nprof = 2
        two = eng.two_streams
        eng.two_streams = False
        for _ in range(nprof):
            eng.step()
        torch.cuda.synchronize()
        eng.two_streams = two
        recs, C.PROFILE = C.PROFILE, None
        fl = sum(r[1] for r in recs)
        ms = sum(r[2].elapsed_time(r[3]) for r in recs)
        by_kind = {}
        for r in recs:
            k = by_kind.setdefault(r[0], [0.0, 0.0, 0])
            k[0] += r[1]; k[1] += r[2].elapsed_time(r[3]); k[2] += 1
        bf = args.precision != 'fp32'
        peak = BF16_MFMA_PEAK_TF if bf else FP32_MFMA_PEAK_TF
        # HBM bytes of the dominant kernel come from separate rocprofv3 --pmc passes (they cannot run inside this
        # process); the committed summary of those passes is reported here, per launch of the named shape.
        traffic, traffic_note = None, None
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r1_conv_pmc.json')
        if bf and os.path.exists(pmc):
            rec = json.load(open(pmc))['launches'][0]
            traffic = round((rec['fetch_bytes'] + rec['write_bytes']) / 1e9, 3)
            traffic_note = ("GB per launch of %s, %s: FETCH_SIZE x2 (gfx950) + WRITE_SIZE from profiles/r1_conv_pmc.json; algorithmic %.3f GB"
                            % (rec['kernel'], rec['shape'], (rec['algorithmic_read_bytes'] + rec['algorithmic_write_bytes']) / 1e9))
        roofline = {"bound": "mfma",
                    "kernel": ("split-bf16 implicit-GEMM family: igemm_patch_bf16x3_kernel (dominant), igemm_nt_bf16x3_kernel, igemm_dma_bf16x3_kernel (3 x v_mfma_f32_32x32x16_bf16 per product block) + exact-fp32 igemm_nt / igemm_wgrad for the Reconstructor"
                               if bf else "igemm_nt_kernel / igemm_wgrad_kernel (fp32 v_mfma_f32_32x32x2_f32)"),
                    "achieved": round(fl / ms / 1e9, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(fl / ms / 1e9 / peak, 4), "traffic": traffic, "traffic_note": traffic_note,
                    "executed_mfma_frac": round({'bf16x3': 3.0, 'f16x2': 2.0}.get(args.precision, 1.0) * fl / ms / 1e9 / peak, 4),
                    "launches_per_step": len(recs) // nprof, "avg_launch_ms": round(ms / len(recs), 4),
                    "conv_ms_per_step": round(ms / nprof, 3), "conv_gflop_per_step": round(fl / nprof / 1e9, 1),
                    "by_kind": {k: {"TFLOP/s": round(v[0] / v[1] / 1e9, 2), "ms_per_step": round(v[1] / nprof, 3),
                                    "launches": v[2] // nprof} for k, v in by_kind.items()},
                    "step_achieved_TFLOPs": round(value / world * GFLOP_PER_IMG / 1e3, 2),
                    "step_frac": round(value / world * GFLOP_PER_IMG / 1e3 / peak, 4)}

    hbm = None
    if rank == 0 and world == 1 and not args.no_roofline:
        try:
            hbm = hbm_subpaths(eng, dev, args.batch)
        except Exception as e:  # noqa: BLE001
            hbm = {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args.size, args.K, args.N, args.cpu_batch, args.cpu_steps, min(os.cpu_count() or 1, args.cpu_threads))
        except Exception as e:  # noqa: BLE001
            cpu = {"error": repr(e)}

    if rank == 0:
        out = {"metric": "training images/sec (warp->G->R->loss) StyleGAN2-%d K=%d" % (args.size, args.K), "value": round(value, 2),
               "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": ("%s generator convs (f16: fp16 operands, 1 MFMA per product, fp32 accumulate; f16x2: fp16 hi+lo weights, 2 MFMAs); reconstructor exact fp32 MFMA forward + weight gradients, split-bf16 input-gradient convs" % args.precision) if args.precision in ('f16', 'f16x2') else ("bf16x3 (generator convs: fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate; reconstructor: %s)" % ("exact fp32 MFMA forward + weight gradients, split-bf16 input-gradient convs" if args.r_precision == 'fp32' else "split-bf16 x3 convs, fp32 wgrad")
                         if args.precision == 'bf16x3' else "fp32 (f32-input MFMA, f32 accumulate)"), "data": "synthetic (random-init weights, z ~ N(0,I))",
               "config": {"workload": "StyleGAN2-FFHQ-%d arch, K=%d, N=%d, ResNet-18 R, batch %d/GPU, %s-space, learn_gammas"
                                      % (args.size, args.K, args.N, args.batch, 'W' if args.w_space else 'Z'),
                          "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                          "algorithmic_gflop_per_image": GFLOP_PER_IMG},
               "last_stats": stats, "roofline": roofline, "hbm_subpaths": hbm, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
