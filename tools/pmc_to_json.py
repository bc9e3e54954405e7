"""Turn the rocprofv3 --pmc passes of tools/pmc_conv16.py (tools/pmc_summary.py's view of them) into profiles/*.json + *.md.
usage: python tools/pmc_to_json.py <pmc dir> <mode> <out prefix>      e.g. gpurun_out/pmc_r2b f16 profiles/r2_conv_pmc"""
import collections, csv, glob, json, sys
d, mode, out = sys.argv[1], sys.argv[2], sys.argv[3]
tab = collections.OrderedDict()
for f in sorted(glob.glob(d + '/*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        kn = r['Kernel_Name']
        if 'igemm' in kn:
            name = 'igemm_' + kn.split('igemm_')[1].split('(')[0]
        elif 'upconv_blur_kernel' in kn:
            name = 'upconv_blur_kernel' + kn.split('upconv_blur_kernel')[1].split('(')[0]
        else:
            continue
        key = (int(r['Dispatch_Id']), name)
        tab.setdefault(key, collections.OrderedDict())
        tab[key][r['Counter_Name']] = tab[key].get(r['Counter_Name'], 0) + float(r['Counter_Value'])
B = 32
shapes = [(512, 512, 64, 'conv %s 512->512 @64x64 9 taps B32'), (256, 256, 128, 'conv %s 256->256 @128x128 9 taps B32'),
          (128, 128, 256, 'conv %s 128->128 @256x256 9 taps B32'),
          (256, 128, 128, 'conv f16 256->128 @128x128 up-conv + blur fused B32'), (256, 128, 128, 'conv f16x2 256->128 @128x128 up-conv + blur fused B32')]
keys = list(tab)
launches = []
lines = ['| launch | kernel | algorithmic HBM bytes (read + write) | FETCH_SIZE x2 + WRITE_SIZE | L2 hit | MFMA-busy share of SIMD cycles | VALU / SALU / LDS instr per MFMA | SQ_WAIT_ANY / WAIT_INST_ANY / ACTIVE (of wave cycles) |', '|---|---|---|---|---|---|---|---|']
for i, (ci, co, h, lab) in enumerate(shapes):
    if 2 * i + 1 >= len(keys):
        break
    c = tab[keys[2 * i + 1]]          # second launch of each shape
    up = 'up-conv' in lab
    ho = 2 * h if up else h
    rd = B * h * h * ci * 4 + co * 9 * ci * 2
    wr = B * ho * ho * co * 4
    fetch, write = c.get('FETCH_SIZE', 0) * 1024 * 2, c.get('WRITE_SIZE', 0) * 1024
    rec = dict(label=(lab % mode) if '%s' in lab else lab, kernel=keys[2 * i + 1][1], shape='%d->%d @%d B=%d' % (ci, co, h, B), fetch_bytes=fetch, write_bytes=write,
               algorithmic_read_bytes=rd, algorithmic_write_bytes=wr, counters=c)
    launches.append(rec)
    mf = c.get('SQ_INSTS_MFMA', 0) or 1
    simd_cycles = c.get('GRBM_GUI_ACTIVE', 0) / 8 * 256 * 4
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    lines.append('| %s | `%s` | %.3f + %.3f GB | %.3f + %.3f GB | %.1f %% | %.1f %% | %.1f / %.1f / %.1f | %.0f %% / %.0f %% / %.0f %% |' % (
        rec['shape'], rec['kernel'], rd / 1e9, wr / 1e9, fetch / 1e9, write / 1e9,
        100 * c.get('TCC_HIT_sum', 0) / max(1, c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0)),
        100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1, simd_cycles),
        c.get('SQ_INSTS_VALU', 0) / mf, c.get('SQ_INSTS_SALU', 0) / mf, c.get('SQ_INSTS_LDS', 0) / mf,
        100 * c.get('SQ_WAIT_ANY', 0) / wc, 100 * c.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc))
json.dump({'mode': mode, 'launches': launches}, open(out + '.json', 'w'), indent=1)
open(out + '_table.md', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
