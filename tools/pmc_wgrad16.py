"""PMC passes of the split-bf16 weight-gradient kernels:  rocprofv3 --kernel-trace --pmc <group> -d DIR/pN -o p -- python tools/pmc_wgrad16.py
   python tools/pmc_wgrad16.py --summarise DIR   prints per kernel: counters of its last dispatch and a few ratios."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from warpedganspace_amd import conv as C
    dev = torch.device('cuda:0'); B = 32
    for ci, co, h, k, s, s2d in [(64, 64, 256, 3, 1, False), (128, 128, 128, 3, 1, False), (64, 128, 256, 3, 2, False), (32, 64, 512, 4, 1, True)]:
        x = torch.randn(B, h, h, ci, device=dev); ho = h // s
        dy = torch.randn(B, ho, ho, co, device=dev)
        dw = torch.zeros(co, k * k, ci, device=dev)
        for prec in ((1, 0) if (k == 3 and s == 1) else (1,)):      # round 5: the direct-fragment kernel in split-bf16 and in exact fp32
            for _ in range(2):
                C.conv2d_wgrad(x, dy, dw, k, stride=s, pad=k // 2, precision=prec)
        torch.cuda.synchronize()


def summarise(d):
    import collections, csv, glob, re
    tab = collections.OrderedDict()
    for f in sorted(glob.glob(d + '/*/*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            mm = re.search(r'(igemm_wgrad16\w*<[^>]*>|wgrad_direct_kernel<[^>]*>)', r['Kernel_Name'])
            if not mm:
                continue
            key = (mm.group(1), r.get('Grid_Size', ''))
            tab.setdefault(key, collections.OrderedDict())
            tab[key][r['Counter_Name']] = float(r['Counter_Value'])      # last dispatch wins
    for (k, g), c in tab.items():
        print(k, 'grid', g)
        for n, v in c.items():
            print('    %-28s %.4g' % (n, v))
        mf = c.get('SQ_INSTS_MFMA') or c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16')
        if mf and 'SQ_INSTS_VALU' in c:
            print('    VALU / MFMA %.1f  LDS / MFMA %.2f  SALU / MFMA %.1f' % (c['SQ_INSTS_VALU'] / mf, c.get('SQ_INSTS_LDS', 0) / mf, c.get('SQ_INSTS_SALU', 0) / mf))
        if 'SQ_BUSY_CYCLES' in c and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
            print('    MFMA busy / SQ busy %.3f; wait_any / wave cycles %.3f; LDS bank conflict / SQ busy %.3f' % (
                c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES'] / 4, c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1), c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_BUSY_CYCLES']))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--summarise':
        summarise(sys.argv[2])
    else:
        run()
