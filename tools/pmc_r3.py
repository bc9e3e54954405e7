"""Round-3 PMC passes of the dominant conv kernels (rocprofv3 --pmc, one counter group per pass, separate from any tracing):
  python tools/pmc_r3.py                      launch every shape below twice (run this under rocprofv3 --kernel-trace --pmc <group>)
  python tools/pmc_r3.py --summarise DIR OUT  read DIR/*/p_counter_collection.csv -> OUT.json + OUT_table.md
FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE is in KB."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

B = 32
# (label, kind, precision, ci, co, h): kind 'conv' = styled stride-1 3x3, 'up' = fused up-sampling layer, 'upT' = transposed conv phases
SHAPES = [('conv fp32w 512->512 @64x64 9 taps B32', 'conv', 'fp32w', 512, 512, 64),
          ('conv fp32w 256->256 @128x128 9 taps B32', 'conv', 'fp32w', 256, 256, 128),
          ('conv fp32w 128->128 @256x256 9 taps B32', 'conv', 'fp32w', 128, 128, 256),
          ('conv fp32 512->512 @64x64 9 taps B32', 'conv', 'fp32', 512, 512, 64),
          ('conv fp32 256->256 @128x128 9 taps B32', 'conv', 'fp32', 256, 256, 128),
          ('conv fp32 128->128 @256x256 9 taps B32', 'conv', 'fp32', 128, 128, 256),
          ('conv fp32 256->128 @128x128 up-conv x4 phases B32', 'upT', 'fp32', 256, 128, 128),
          ('conv f16 256->256 @128x128 9 taps B32', 'conv', 'f16', 256, 256, 128),
          ('conv f16 128->128 @256x256 9 taps B32', 'conv', 'f16', 128, 128, 256),
          ('conv f16x2 256->128 @128x128 up-conv + blur fused B32', 'up', 'f16x2', 256, 128, 128),
          ('conv f16x2 512->256 @64x64 up-conv + blur fused B32', 'up', 'f16x2', 512, 256, 64),
          # round 4: producer-written fp16 plane through the patch kernel's XF16 form, and the few-channel halo kernel (StyleGAN2-1024, B = 8)
          ('conv f16 128->128 @256x256 9 taps B32 (fp16 plane)', 'plane', 'f16', 128, 128, 256),
          ('conv f16x2 32->32 @1024x1024 9 taps B8', 'halo', 'f16x2', 32, 32, 1024),
          ('conv f16x2 64->64 @512x512 9 taps B8', 'halo', 'f16x2', 64, 64, 512),
          # round 5: the fused up-sampling kernel in split-bf16 (StyleGAN2-256's 32 -> 64 layer under the default policy)
          ('conv bf16x3 512->512 @32x32 up-conv + blur fused B32', 'up', 'bf16x3', 512, 512, 32),
          # round 6: the split-bf16 F(2,3) kernel (conv_wino_bf16.hip) and the direct split-bf16 patch kernel it replaces
          ('conv bf16x3w 512->512 @64x64 9 taps B32', 'conv', 'bf16x3w', 512, 512, 64),
          ('conv bf16x3w 256->256 @128x128 9 taps B32', 'conv', 'bf16x3w', 256, 256, 128),
          ('conv bf16x3w 128->128 @256x256 9 taps B32', 'conv', 'bf16x3w', 128, 128, 256),
          ('conv bf16x3 512->512 @64x64 9 taps B32', 'conv', 'bf16x3', 512, 512, 64),
          ('conv bf16x3 256->256 @128x128 9 taps B32', 'conv', 'bf16x3', 256, 256, 128)]
if os.environ.get('PMC_ONLY'):       # restrict a run AND its summary to the shapes whose label contains this text (tools/pmc_quick.sh)
    SHAPES = [s_ for s_ in SHAPES if os.environ['PMC_ONLY'] in s_[0]]


def run():
    import torch
    from warpedganspace_amd import conv as C
    dev = torch.device('cuda:0')
    for label, kind, prec, ci, co, h in SHAPES:
        m = C.precision_code(prec)
        B = 8 if kind == 'halo' else 32
        x = torch.randn(B, h, h, ci, device=dev); w = torch.randn(co, 9, ci, device=dev) / (9 * ci) ** 0.5
        s = torch.randn(B, ci, device=dev); dm = torch.rand(B, co, device=dev)
        ws = C.SplitCache(w) if m in (C.FP32W, C.BF16W) else (C.split_weight(w, m) if m else None)
        if kind in ('conv', 'halo'):
            y = torch.empty(B, h, h, co, device=dev)
            am = x.abs().amax().reshape(1)
            fn = lambda: C.conv2d(x, w, 3, pad=1, out=y, a_scale=s, col_scale=dm, act_slope=0.2, gain=1.41, precision=m, w_split=ws, a_amax=am)
        elif kind == 'plane':
            y = torch.empty(B, h, h, co, device=dev)
            am = torch.full((1,), 4.0, device=dev)
            xp = (x * 512.0).half().view(torch.int16)
            fn = lambda: C.conv2d(xp, w, 3, pad=1, out=y, col_scale=dm, act_slope=0.2, gain=1.41, precision=m, w_split=ws, a_amax=am, x_f16=True)
        elif kind == 'upT':
            t = torch.empty(B, 2 * h + 1, 2 * h + 1, co, device=dev)
            fn = lambda: C.conv_transpose2d_s2(x, w, out=t, a_scale=s, col_scale=dm, precision=m, w_split=ws)
        else:
            k1 = torch.tensor([1., 3., 3., 1.]); kern = (k1[:, None] * k1[None, :] / 64 * 4).to(dev)
            nz, nw, bias = torch.randn(4 * h * h, device=dev), torch.ones(1, device=dev), torch.zeros(co, device=dev)
            fn = lambda: C.upconv_blur_act(x, ws, kern, s, ci, dm, nz, nw, bias, m)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        print(label, flush=True)


def summarise(d, out):
    import collections, csv, glob, json, re
    tab = collections.OrderedDict()
    for f in sorted(glob.glob(d + '/*/p_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name']
            mm = re.search(r'(igemm_\w+<[^>]*>|patch_dma_kernel<[^>]*>|upconv_blur_kernel<[^>]*>|wino_f32_kernel<[^>]*>|wino16_kernel<[^>]*>|halo3x3_kernel<[^>]*>)', kn)
            if not mm:
                continue
            key = (int(r['Dispatch_Id']), mm.group(1))
            tab.setdefault(key, collections.OrderedDict())
            tab[key][r['Counter_Name']] = tab[key].get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    # dispatches in launch order, two per shape: take the second
    keys = sorted(tab)
    recs, lines = [], ['| launch | kernel | algorithmic HBM bytes (read + write) | FETCH_SIZE x2 + WRITE_SIZE | L2 hit | MFMA-busy share of SIMD cycles | VALU / SALU / LDS instr per MFMA | SQ_WAIT_ANY / WAIT_INST_ANY / ACTIVE (of wave cycles) |', '|---|---|---|---|---|---|---|---|']
    for i, (label, kind, prec, ci, co, h) in enumerate(SHAPES):
        if 2 * i + 1 >= len(keys):
            break
        sym, c = keys[2 * i + 1][1], tab[keys[2 * i + 1]]
        B = 8 if kind == 'halo' else 32
        ho = 2 * h if kind == 'up' else (2 * h + 1 if kind == 'upT' else h)
        rd = B * h * h * ci * (2 if kind == 'plane' else 4) + co * ci * (64 if prec == 'fp32w' else (36 if prec == 'fp32' else 18))
        wr = B * ho * ho * co * 4
        fetch, write = c.get('FETCH_SIZE', 0) * 1024 * 2, c.get('WRITE_SIZE', 0) * 1024
        mf = c.get('SQ_INSTS_MFMA', 0) or 1
        simd = c.get('GRBM_GUI_ACTIVE', 0) / 8 * 256 * 4
        wc = c.get('SQ_WAVE_CYCLES', 0) or 1
        recs.append(dict(symbol=sym, shape=label, hbm_bytes_per_launch=fetch + write, algorithmic_bytes_per_launch=rd + wr, fetch_bytes=fetch,
                         write_bytes=write, mfma_busy_frac=c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1, simd), counters=c))
        lines.append('| %s | `%s` | %.3f + %.3f GB | %.3f + %.3f GB | %.1f %% | %.1f %% | %.1f / %.1f / %.1f | %.0f %% / %.0f %% / %.0f %% |' % (
            label, sym, rd / 1e9, wr / 1e9, fetch / 1e9, write / 1e9,
            100 * c.get('TCC_HIT_sum', 0) / max(1, c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0)),
            100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1, simd),
            c.get('SQ_INSTS_VALU', 0) / mf, c.get('SQ_INSTS_SALU', 0) / mf, c.get('SQ_INSTS_LDS', 0) / mf,
            100 * c.get('SQ_WAIT_ANY', 0) / wc, 100 * c.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc))
    json.dump({'kernels': recs}, open(out + '.json', 'w'), indent=1)
    open(out + '_table.md', 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--summarise':
        summarise(sys.argv[2], sys.argv[3])
    else:
        run()
