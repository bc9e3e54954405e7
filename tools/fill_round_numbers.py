#!/usr/bin/env python
"""Numbers of DESIGN.md / README.md that come from the published round profiles (profiles/r6_*).
  python tools/fill_round_numbers.py "<suite text>"                     fill the R6_* placeholders
  python tools/fill_round_numbers.py --from-rev REV "<suite text>"      the placeholders were filled from REV's profiles: replace every string
                                                                        REV's profiles gave by the one the working tree's profiles give"""
import io, json, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = sys.argv[1:]
FROM_REV = None
if '--from-rev' in argv:
    i = argv.index('--from-rev'); FROM_REV = argv[i + 1]; del argv[i:i + 2]
suite = argv[0] if argv else None


def _open(name, rev):
    if rev:
        return io.StringIO(subprocess.run(['git', 'show', '%s:profiles/%s' % (rev, name)], cwd=root, capture_output=True, text=True, check=True).stdout)
    return open(os.path.join(root, 'profiles', name))


def sp(x, nd=1):          # 1297.99 -> '1 298.0'
    s = ('%.' + str(nd) + 'f') % x
    i, f = s.split('.') if nd else (s, '')
    i = re.sub(r'(?<=\d)(?=(\d{3})+$)', ' ', i)
    return i + ('.' + f if nd else '')


def compute(rev=None):
    d = json.load(_open('r6_bench_default.json', rev))
    p, u = d['product'], d['product']['uncalibrated']
    pc, upc = p['precision_check'], u['precision_check']
    ph = {}
    for ln in _open('r6_phases_auto.md', rev):
        m = re.match(r'\| ([^|]+) \| (\d+) \| ([\d.]+) \|', ln)
        if m:
            ph[m.group(1).strip()] = float(m.group(3))
        m = re.match(r'one step: (\d+) kernels, ([\d.]+) ms from first start to last end, ([\d.]+) ms of kernel time', ln)
        if m:
            ph['sum'] = float(m.group(3))
    ab = {}
    for ln in _open('r6_ab_schedule.txt', rev):
        if ln.startswith('{'):
            e = json.loads(ln); ab[e['variant']] = min(e['ms_per_step'])
    w16 = 0.0
    for ln in _open('r6_step_auto_kernel_stats.md', rev):
        if 'wino16_kernel<true, true>' in ln:
            w16 = float(ln.split('|')[-2])
    o = d['others_images_per_sec']
    others = ('`bf16x3` %s, `bf16x3w` %s, `f16` %s (reported: over the gate), `f16x2` %s, W-space %s, `[R fp32]` %s; cfg2 ProgGAN-1024 B=32 **%s** (ProgGAN-256 %s); '
              'cfg4 BigGAN-128 %s, BigGAN-256 %s; cfg5 StyleGAN2-1024 K=200 N=64 B=8: `fp32w` %s, fp32 %s, **`auto` %s**, `bf16x3` %s' % (
                  sp(o['cfg3_bf16x3'], 0), sp(o['cfg3_bf16x3w'], 0), sp(o['cfg3_f16'], 0), sp(o['cfg3_f16x2'], 0), sp(o['cfg3_auto_Wspace'], 0), sp(o['cfg3_auto_Rfp32'], 0),
                  sp(o['cfg2_proggan1024_auto'], 0), sp(o['cfg2_proggan256_auto'], 0), sp(o['cfg4_biggan128_auto'], 0), sp(o['cfg4_biggan256_auto'], 0),
                  sp(o['cfg5_sg1024_fp32w'], 0), sp(o['cfg5_sg1024_fp32'], 0), sp(o['cfg5_sg1024_auto'], 0), sp(o['cfg5_sg1024_bf16x3'], 0)))
    r = d['roofline']
    return {
        # key: (string, context regex with ONE group around the string — used by --from-rev where the bare string is too short to be unique)
        'R6_HEAD_STEP': ('%.3f' % r['step_frac'], r'step executes (\S+) of the peak'),
        'R6_HEAD_FRAC': ('%.3f' % r['frac'], None), 'R6_HEAD_TF': ('%.1f TF executed' % r['achieved'], None),
        'R6_HEAD': (sp(d['value']), None), 'R6_PROD_MS': ('%.2f' % p['ms_per_step'], None), 'R6_PROD': (sp(p['value']), None), 'R6_TABLE': (p['table'], None),
        'R6_IMAX': ('%.2e' % pc['image_max'], None), 'R6_UNCAL_IMAX': ('%.1e' % upc['image_max'], None),
        'R6_UNCAL_OVER': ('%.1f %%' % (100 * upc['over_gate_frac']), r'reported: [\d ]+\.\d \([\de.+-]+, ([\d.]+ %) over\)'),
        'R6_UNCAL': (sp(u['value']), None), 'R6_DIRECT': (sp(d['direct_fp32']['value']), None),
        'R6_W16_MS': ('%.2f ms per step in eight' % w16, None), 'R6_W16_TF': ('%.0f TF\n' % p['roofline']['achieved'], None),
        'R6_W16_FRAC': ('%.3f of the 2.5 PF' % p['roofline']['frac'], None),
        'R6_PH_SUM': ('%.1f ms of kernel time' % ph.get('sum', 0), None), 'R6_PH_GZ': ('G(z) forward %.2f' % ph.get('G(z) forward (nothing saved)', 0), None),
        'R6_PH_GS': ('forward %.2f (each' % ph.get('RBF warp + G(z+shift) forward', 0), None),
        'R6_PH_RF': ('Reconstructor forward %.2f' % ph.get('Reconstructor forward', 0), None), 'R6_PH_RB': ('+ backward %.2f' % ph.get('loss + Reconstructor backward', 0), None),
        'R6_PH_GB': ('backward %.2f (plain' % ph.get('G backward (input gradient)', 0), None),
        'R6_AB_CHAIN': ('not training) %.1f ms' % ab.get('chain only (static un-shifted batch: NOT training)', 0), None), 'R6_AB_BASE': ('against %.1f' % ab.get('baseline', 0), None),
        'R6_OTHERS': (others, None), 'R6_LAUNCHES': ('%.0f library launches' % d['host']['library_launches_per_step'], None),
        'R6_ENQ': ('%.1f ms to enqueue' % d['host']['host_enqueue_ms_per_step'], None),
        'R6_CFG5': ('cfg5 265 → %s' % sp(o['cfg5_sg1024_auto'], 0), None), 'R6_CFG2': ('cfg2 375 → %s' % sp(o['cfg2_proggan1024_auto'], 0), None),
    }


new = compute(None)
for fn in ('DESIGN.md', 'README.md'):
    path = os.path.join(root, fn)
    s = open(path).read()
    if FROM_REV:
        old = compute(FROM_REV)
        for k in sorted(new, key=len, reverse=True):
            (so, ctx), (sn, _) = old[k], new[k]
            if so == sn:
                continue
            if ctx:
                s = re.sub(ctx, lambda m: m.group(0).replace(m.group(1), sn if m.group(1) == so else m.group(1)), s)
            elif len(so) >= 5:
                s = s.replace(so, sn)
            else:
                print('  (not replaced, too short to be unique: %s %r -> %r)' % (k, so, sn))
    else:
        # (the placeholder sits where the NUMBER goes: the context words of the strings above are already in the text)
        for k in sorted(new, key=len, reverse=True):
            num = re.search(r'[\d][\d .e+-]*[\d%]|`[^`]+`.*', new[k][0])
            s = s.replace(k, new[k][0] if k in ('R6_OTHERS', 'R6_TABLE') else (num.group(0).strip() if num else new[k][0]))
        if suite:
            s = s.replace('R6_SUITE_S', suite).replace('R6_SUITE', suite)
    left = sorted(set(re.findall(r'R6_[A-Z0-9_]+', s)))
    open(path, 'w').write(s)
    print(fn, 'updated;', 'placeholders left:', left)
