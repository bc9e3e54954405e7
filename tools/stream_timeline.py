"""Timeline of one multi-stream training step from a rocprofv3 --kernel-trace run: per HIP queue, the spans of the step's phases and
how much of the wall time each queue is busy.  usage: python tools/stream_timeline.py <dir-with-*_kernel_trace.csv>"""
import csv, glob, re, sys, collections
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); n = re.sub(r'\(.*$', '', n)
    return n[:60]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?')) for r in rows), key=lambda e: e[0])
adam = [i for i, e in enumerate(ev) if e[2].startswith('adam_kernel')]
ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] != i + 1]
a, b = ends[-4] + 1, ends[-3] + 1
t0, t1 = ev[ends[-4]][1], ev[ends[-3]][1]
print('step wall (adam end -> adam end): %.2f ms' % ((t1 - t0) / 1e6))
step = [e for e in ev if e[1] > t0 and e[0] < t1]
byq = collections.defaultdict(list)
for e in step:
    byq[e[3]].append(e)
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
print('any queue busy: %.2f ms; sum of kernel durations: %.2f ms' % (union([(max(s, t0), min(e, t1)) for s, e, _, _ in step]) / 1e6, sum(min(e, t1) - max(s, t0) for s, e, _, _ in step) / 1e6))
for q, es in sorted(byq.items(), key=lambda kv: kv[1][0][0]):
    print('\nqueue %s: %d kernels, busy %.2f ms, from +%.2f to +%.2f ms' % (q, len(es), union([(max(s, t0), min(e, t1)) for s, e, _, _ in es]) / 1e6, (es[0][0] - t0) / 1e6, (es[-1][1] - t0) / 1e6))
    # coarse phases: consecutive runs, report every 2 ms window's dominant kernel
    win = collections.OrderedDict()
    for s, e, n, _ in es:
        w = int((s - t0) / 2e6)
        win.setdefault(w, collections.Counter())[n] += (e - s) / 1e6
    for w, c in win.items():
        top = c.most_common(2)
        print('   +%2d..%2d ms: %s' % (2 * w, 2 * w + 2, '; '.join('%s %.2f' % (n, t) for n, t in top)))
