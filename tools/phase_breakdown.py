"""Split one steady-state single-stream training step of a rocprofv3 --kernel-trace run into its phases and list where each phase's
time goes.  usage: python tools/phase_breakdown.py <dir-with-*_kernel_trace.csv> [n_top]
Phases by marker kernels of the step (trainer.TrainStep.step, single stream):
  G(z) forward | RBF warp | G(z+shift) forward | Reconstructor forward | loss | Reconstructor backward | G backward | RBF backward | Adam"""
import csv, glob, re, sys, collections

d = sys.argv[1]; ntop = int(sys.argv[2]) if len(sys.argv) > 2 else 6
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); n = re.sub(r'\(.*$', '', n)
    return n[:70]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows), key=lambda e: e[0])
# steps = spans between consecutive adam_kernel groups; take the LAST complete one before the profiled (event-timed) steps is fine: any
adam = [i for i, e in enumerate(ev) if e[2].startswith('adam_kernel')]
ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] != i + 1]      # last adam launch of each step
if len(ends) < 4:
    raise SystemExit('need >= 4 steps in the trace')
a, b = ends[-3] + 1, ends[-2] + 1          # one full step
step = ev[a:b]
def first(pred, start=0):
    for i in range(start, len(step)):
        if pred(step[i][2]):
            return i
    return len(step)
i_rbf = first(lambda n: n.startswith('rbf_fwd'))
i_gs_end = first(lambda n: n.startswith('pack_pair_kernel'), i_rbf)   # R's input packing (the two images -> one NHWC tensor)
i_loss = first(lambda n: n.startswith('loss_rows_kernel'))
i_gb = first(lambda n: n.startswith('unpack_pair'), i_loss) + 1      # R's input gradient -> d image: the generator's backward follows
i_rbfb = first(lambda n: n.startswith('rbf_bwd'), i_loss)
i_adam = first(lambda n: n.startswith('adam_kernel'), i_loss)
# R forward starts at the first kernel after the last ToRGB of the second generator pass
tor = [i for i in range(i_rbf, i_loss) if step[i][2].startswith('torgb')]
i_rf = i_gs_end if i_gs_end < len(step) else ((tor[-1] + 1) if tor else i_rbf)
cuts = [('G(z) forward (nothing saved)', 0, i_rbf), ('RBF warp + G(z+shift) forward', i_rbf, i_rf), ('Reconstructor forward', i_rf, i_loss),
        ('loss + Reconstructor backward', i_loss, i_gb), ('G backward (input gradient)', i_gb, i_rbfb), ('RBF backward + Adam', i_rbfb, len(step))]
wall = (step[-1][1] - step[0][0]) / 1e6
print('one step: %d kernels, %.2f ms from first start to last end, %.2f ms of kernel time' % (len(step), wall, sum(e[1] - e[0] for e in step) / 1e6))
print('\n| phase | kernels | kernel ms | span ms | top kernels (ms, calls) |\n|---|---|---|---|---|')
for name, lo, hi in cuts:
    seg = step[lo:hi]
    if not seg:
        continue
    t = collections.defaultdict(lambda: [0.0, 0])
    for s, e, n in seg:
        t[n][0] += (e - s) / 1e6; t[n][1] += 1
    top = sorted(t.items(), key=lambda kv: -kv[1][0])[:ntop]
    print('| %s | %d | %.2f | %.2f | %s |' % (name, len(seg), sum(e - s for s, e, _ in seg) / 1e6, (seg[-1][1] - seg[0][0]) / 1e6,
                                            '; '.join('`%s` %.2f (%d)' % (n, v[0], v[1]) for n, v in top)))
