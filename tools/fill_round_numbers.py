#!/usr/bin/env python
"""Fill the R6_* placeholders of DESIGN.md / README.md from the published round profiles (profiles/r6_*).
usage: python tools/fill_round_numbers.py [suite-text]   e.g. "514 passed, 2 skipped in 512 s (gpurun_out/r6c_tests.log)" """
import json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pr = os.path.join(root, 'profiles')
d = json.load(open(os.path.join(pr, 'r6_bench_default.json')))
p, u = d['product'], d['product']['uncalibrated']
pc, upc = p['precision_check'], u['precision_check']
def sp(x, nd=1):          # 1 297.99 -> '1 298.0'
    s = ('%.' + str(nd) + 'f') % x
    i, f = s.split('.') if nd else (s, '')
    i = re.sub(r'(?<=\d)(?=(\d{3})+$)', ' ', i)
    return i + ('.' + f if nd else '')
ph = {}
for ln in open(os.path.join(pr, 'r6_phases_auto.md')):
    m = re.match(r'\| ([^|]+) \| (\d+) \| ([\d.]+) \|', ln)
    if m:
        ph[m.group(1).strip()] = float(m.group(3))
    m = re.match(r'one step: (\d+) kernels, ([\d.]+) ms from first start to last end, ([\d.]+) ms of kernel time', ln)
    if m:
        ph['sum'] = float(m.group(3))
ab = {}
f = os.path.join(pr, 'r6_ab_schedule.txt')
if os.path.exists(f):
    for ln in open(f):
        if ln.startswith('{'):
            e = json.loads(ln); ab[e['variant']] = min(e['ms_per_step'])
w16 = None
for ln in open(os.path.join(pr, 'r6_step_auto_kernel_stats.md')):
    if 'wino16_kernel<true, true>' in ln:
        w16 = float(ln.split('|')[-2])
o = d['others_images_per_sec']
others = ('`bf16x3` %s, `bf16x3w` %s, `f16` %s (reported: over the gate), `f16x2` %s, W-space %s, `[R fp32]` %s; cfg2 ProgGAN-1024 B=32 **%s** (ProgGAN-256 %s); '
          'cfg4 BigGAN-128 %s, BigGAN-256 %s; cfg5 StyleGAN2-1024 K=200 N=64 B=8: `fp32w` %s, fp32 %s, **`auto` %s**, `bf16x3` %s' % (
              sp(o['cfg3_bf16x3'], 0), sp(o['cfg3_bf16x3w'], 0), sp(o['cfg3_f16'], 0), sp(o['cfg3_f16x2'], 0), sp(o['cfg3_auto_Wspace'], 0), sp(o['cfg3_auto_Rfp32'], 0),
              sp(o['cfg2_proggan1024_auto'], 0), sp(o['cfg2_proggan256_auto'], 0), sp(o['cfg4_biggan128_auto'], 0), sp(o['cfg4_biggan256_auto'], 0),
              sp(o['cfg5_sg1024_fp32w'], 0), sp(o['cfg5_sg1024_fp32'], 0), sp(o['cfg5_sg1024_auto'], 0), sp(o['cfg5_sg1024_bf16x3'], 0)))
suite = sys.argv[1] if len(sys.argv) > 1 else 'see gpurun_out'
vals = {
    'R6_HEAD_STEP': '%.3f' % d['roofline']['step_frac'], 'R6_HEAD_FRAC': '%.3f' % d['roofline']['frac'], 'R6_HEAD_TF': '%.1f' % d['roofline']['achieved'],
    'R6_HEAD': '%s' % sp(d['value']), 'R6_PROD_MS': '%.2f' % p['ms_per_step'], 'R6_PROD': sp(p['value']), 'R6_TABLE': p['table'],
    'R6_IMAX': '%.2e' % pc['image_max'], 'R6_UNCAL_IMAX': '%.1e' % upc['image_max'], 'R6_UNCAL_OVER': '%.1f %%' % (100 * upc['over_gate_frac']),
    'R6_UNCAL': sp(u['value']), 'R6_DIRECT': sp(d['direct_fp32']['value']),
    'R6_W16_MS': '%.2f' % (w16 or 0), 'R6_W16_TF': '%.0f' % p['roofline']['achieved'], 'R6_W16_FRAC': '%.3f' % p['roofline']['frac'],
    'R6_PH_SUM': '%.1f' % ph.get('sum', 0), 'R6_PH_GZ': '%.2f' % ph.get('G(z) forward (nothing saved)', 0), 'R6_PH_GS': '%.2f' % ph.get('RBF warp + G(z+shift) forward', 0),
    'R6_PH_RF': '%.2f' % ph.get('Reconstructor forward', 0), 'R6_PH_RB': '%.2f' % ph.get('loss + Reconstructor backward', 0), 'R6_PH_GB': '%.2f' % ph.get('G backward (input gradient)', 0),
    'R6_AB_CHAIN': '%.1f' % ab.get('chain only (static un-shifted batch: NOT training)', 0), 'R6_AB_BASE': '%.1f' % ab.get('baseline', 0),
    'R6_OTHERS': others, 'R6_LAUNCHES': '%.0f' % d['host']['library_launches_per_step'], 'R6_ENQ': '%.1f' % d['host']['host_enqueue_ms_per_step'],
    'R6_SUITE_S': suite, 'R6_SUITE': suite, 'R6_CFG5': sp(o['cfg5_sg1024_auto'], 0), 'R6_CFG2': sp(o['cfg2_proggan1024_auto'], 0),
}
for fn in ('DESIGN.md', 'README.md'):
    path = os.path.join(root, fn)
    s = open(path).read()
    for k in sorted(vals, key=len, reverse=True):
        s = s.replace(k, vals[k])
    left = sorted(set(re.findall(r'R6_[A-Z0-9_]+', s)))
    open(path, 'w').write(s)
    print(fn, 'filled;', 'left over:', left)
