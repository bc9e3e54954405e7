"""Same-process, same-box A/B of the prefetched pass's schedule (trainer.TrainStep: split / tail stages).

  python tools/ab_tail.py [--precision fp32w,mixed] [--steps 60] [--rounds 2]

One engine per arithmetic; the variants are attribute settings of that engine, timed round-robin (`rounds` times each, so that a drift
of the box shows up as a spread instead of as a difference).  Prints one JSON line per (precision, variant)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402

VARIANTS = [
    ('tail off', dict(tail_prefetch=False)),
    ('tail default/16', dict(tail_prefetch=True, tail_pause_res=None, tail_hook_res=16)),
    ('tail default/32', dict(tail_prefetch=True, tail_pause_res=None, tail_hook_res=32)),
    ('tail 128/16', dict(tail_prefetch=True, tail_pause_res=128, tail_hook_res=16)),
    ('tail 256/16', dict(tail_prefetch=True, tail_pause_res=256, tail_hook_res=16)),
    ('weights inline', dict(prepare_wt=False)),
    ('weights ahead', dict(prepare_wt=True)),
    ('split 16', dict(split_pause_res=16)),
    ('split 32', dict(split_pause_res=32)),
    ('split 64', dict(split_pause_res=64)),
    ('torch sampler', dict(torch_sampler=True)),
    ('kernel sampler', dict(torch_sampler=False)),
    # round 5: module-level routing switches ('conv.<NAME>' keys are set on warpedganspace_amd.conv instead of the engine)
    ('wgrad staged', {'conv.WGRAD_DIRECT': False}),
    ('wgrad direct', {'conv.WGRAD_DIRECT': True}),
    ('up bf16 unfused', {'conv.UPCONV_FUSED_MIN_H': {1: 1 << 30, 2: 16, 3: 16}}),
    ('up bf16 fused', {'conv.UPCONV_FUSED_MIN_H': {1: 32, 2: 16, 3: 16}}),
    ('BN stats pass', {'rr.BN_EPILOGUE_STATS': False, 'rr.BN_FUSED_APPLY': False}),
    ('BN stats in epilogue', {'rr.BN_EPILOGUE_STATS': True, 'rr.BN_FUSED_APPLY': False}),
    ('BN stats in epilogue + fused apply', {'rr.BN_EPILOGUE_STATS': True, 'rr.BN_FUSED_APPLY': True}),
    ('R +0 launches', {'bn.debug_extra_launches': 0}),
    ('R +40 empty launches', {'bn.debug_extra_launches': 2}),
    ('R +80 empty launches', {'bn.debug_extra_launches': 4}),
    # library development flags ('env.<NAME>': set / unset in the environment, then wgs_dev_reload_flags())
    ('patch regs', {'env.WGS_PATCH_NODMA': '1', 'env.WGS_PATCH_DMA_BM': None}),
    ('patch dma 128', {'env.WGS_PATCH_NODMA': None, 'env.WGS_PATCH_DMA_BM': '128'}),
    ('patch dma 256', {'env.WGS_PATCH_NODMA': None, 'env.WGS_PATCH_DMA_BM': '256'}),
    ('baseline', dict(debug_static_unshifted=False, mid_after_r=False)),
    ('chain only (static un-shifted batch: NOT training)', dict(debug_static_unshifted=True, mid_after_r=False)),
    ('mid stage behind R', dict(debug_static_unshifted=False, mid_after_r=True)),
    ('baseline again', dict(debug_static_unshifted=False, mid_after_r=False)),
]
CONFIGS = {'cfg3': ('stylegan2', 128, 32, 32, 256), 'cfg5': ('stylegan2', 200, 64, 8, 1024), 'cfg2': ('proggan', 64, 16, 32, 1024), 'cfg4': ('biggan', 128, 32, 16, 128)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--precision', default='fp32w,mixed')
    ap.add_argument('--config', default='cfg3', choices=tuple(CONFIGS))
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--only', default='', help='comma-separated variant names (default: all)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    if os.environ.get('WGS_RGB') == '0':            # development A/B: ToRGB as its own launch everywhere
        from warpedganspace_amd import conv as _C
        _C.RGB_FUSED = False
    variants = [v for v in VARIANTS if not args.only or v[0] in args.only.split(',')]
    for prec in args.precision.split(','):
        gan, K, N, B, size = CONFIGS[args.config]
        eng = bench.build(dev, gan, K, N, B, precision=prec, size=size)
        for _ in range(10):
            eng.step()
        torch.cuda.synchronize()
        res = {name: [] for name, _ in variants}
        for _ in range(args.rounds):
            for name, kw in variants:
                for k, v in kw.items():
                    if k.startswith('rr.'):
                        from warpedganspace_amd import reconstructor as _RR2
                        setattr(_RR2, k[3:], v)
                    elif k.startswith('bn.'):
                        from warpedganspace_amd import reconstructor as _RR
                        setattr(_RR._BN, k[3:], v)
                    elif k.startswith('conv.'):
                        from warpedganspace_amd import conv as _CC
                        setattr(_CC, k[5:], v)
                    elif k.startswith('env.'):
                        from warpedganspace_amd import _lib as _LL
                        if v is None:
                            os.environ.pop(k[4:], None)
                        else:
                            os.environ[k[4:]] = v
                        _LL.lib().wgs_dev_reload_flags()
                    else:
                        setattr(eng, k, v)
                for _ in range(args.warmup):
                    eng.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    eng.step()
                torch.cuda.synchronize()
                res[name].append(1e3 * (time.perf_counter() - t0) / args.steps)
        for name, _ in variants:
            print(json.dumps({'precision': prec, 'variant': name, 'ms_per_step': [round(v, 3) for v in res[name]],
                              'config': args.config, 'img_per_s': round(B * 1e3 / min(res[name]), 1)}), flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
