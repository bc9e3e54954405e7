"""One multi-stream training step from a rocprofv3 --kernel-trace run as a compact event list: per kernel (start offset us, duration us,
queue, short name), consecutive launches of one kernel name on one queue merged.  usage: python tools/tl_dump.py <dir>"""
import csv, glob, re, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); n = re.sub(r'\(.*$', '', n)
    return n[:48]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?')) for r in rows), key=lambda e: e[0])
adam = [i for i, e in enumerate(ev) if e[2].startswith('adam_kernel')]
ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] != i + 1]
t0, t1 = ev[ends[-4]][1], ev[ends[-3]][1]
step = [e for e in ev if e[1] > t0 and e[0] < t1]
qs = sorted(set(e[3] for e in step))
print('step %.3f ms, queues %s' % ((t1 - t0) / 1e6, qs))
out = []
for s, e, n, q in step:
    if out and out[-1][3] == q and out[-1][2] == n and s - out[-1][1] < 20000:
        out[-1][1] = e; out[-1][4] += 1; out[-1][5] += e - s
    else:
        out.append([s, e, n, q, 1, e - s])
for s, e, n, q, c, busy in out:
    print('%8.1f %8.1f q%s %-48s x%d busy %.1f' % ((s - t0) / 1e3, (e - s) / 1e3, qs.index(q), n, c, busy / 1e3))
