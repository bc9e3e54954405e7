"""PMC passes of the Winograd fp32 kernel vs the direct fp32 kernel on one shape.
  python tools/pmc_wino.py [ci co h]              launch both twice (run under rocprofv3 --kernel-trace --pmc <group>)
  python tools/pmc_wino.py --summarise DIR        read DIR/*/p_counter_collection.csv, print counters per kernel (second dispatch)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(ci=512, co=512, h=64, B=32):
    import torch
    from warpedganspace_amd import conv as C
    dev = torch.device('cuda:0')
    x = torch.randn(B, h, h, ci, device=dev)
    w = C.pack_weight(torch.randn(co, ci, 3, 3, device=dev) / (9 * ci) ** 0.5)
    cache = C.SplitCache(w)
    s, dm = torch.randn(B, ci, device=dev), torch.rand(B, co, device=dev)
    nz, nw, bias = torch.randn(h * h, device=dev), torch.ones(1, device=dev), torch.zeros(co, device=dev)
    y = torch.empty(B, h, h, co, device=dev)
    epi = dict(a_scale=s, col_scale=dm, bias=bias, noise=nz, noise_w=nw, act_slope=0.2, gain=1.41, out=y, w_split=cache)
    for prec in (C.FP32W, 0):
        for _ in range(2):
            C.conv2d(x, w, 3, pad=1, precision=prec, **epi)
        torch.cuda.synchronize()


def summarise(d):
    import collections, csv, glob
    tab = collections.OrderedDict()
    for f in sorted(glob.glob(d + '/*/p_counter_collection.csv')):
        seen = collections.Counter()
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name']
            if 'wino_f32' not in kn and 'igemm_nt16' not in kn:
                continue
            key = 'wino' if 'wino_f32' in kn else 'direct'
            did = int(r['Dispatch_Id'])
            tab.setdefault(key, {}).setdefault(did, {})
            tab[key][did][r['Counter_Name']] = tab[key][did].get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    for key, disp in tab.items():
        # merge counters over passes: dispatch ids repeat per pass in the same order -> take the LAST dispatch of each pass
        merged = {}
        for did in sorted(disp):
            merged.update(disp[did])
        print(key)
        for k, v in merged.items():
            print('   %-32s %.4g' % (k, v))
        wc = merged.get('SQ_WAVE_CYCLES', 0) or 1
        simd = merged.get('GRBM_GUI_ACTIVE', 0) / 8 * 256 * 4
        if simd:
            print('   MFMA busy / SIMD cycles: %.1f %%' % (100 * merged.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd))
        for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_INST_CYCLES_VMEM', 'SQ_ACTIVE_INST_VMEM'):
            if k in merged:
                print('   %s / wave cycles: %.1f %%' % (k, 100 * merged[k] / wc))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--summarise':
        summarise(sys.argv[2])
    else:
        run(*[int(a) for a in sys.argv[1:4]])
