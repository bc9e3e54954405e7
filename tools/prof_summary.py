"""Summarise a rocprofv3 `--kernel-trace --stats --output-format csv` run as a markdown table.
usage: python tools/prof_summary.py <dir-with-*_kernel_stats.csv> [steps] [--steps-only]
--steps-only: read *_kernel_trace.csv instead and count only the kernels of complete training steps (from the launch after one step's last
adam_kernel to the next step's last adam_kernel) — the calibration and precision check of an `auto` engine run hundreds of generator passes in
front of the first step and would drown the table."""
import collections, csv, glob, re, sys
steps_only = '--steps-only' in sys.argv
argv = [a for a in sys.argv if a != '--steps-only']
d = argv[1]; steps = float(argv[2]) if len(argv) > 2 else None
if steps_only:
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))), key=lambda e: e[0])
    adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
    ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] != i + 1]      # last adam launch of each step
    # a step's first kernel follows the previous step's last adam launch; the FIRST step of a run has the calibration in front of it: dropped
    agg = collections.defaultdict(lambda: [0, 0])
    for a, b in zip(ends[:-1], ends[1:]):
        if ev[b][0] - ev[a][1] > 0.5e9:          # (a gap of > 0.5 s between two steps: something else ran there, e.g. a precision check)
            continue
        for s_, e_, n in ev[a + 1:b + 1]:
            agg[n][0] += e_ - s_; agg[n][1] += 1
        steps_counted = locals().get('steps_counted', 0) + 1
    steps = float(steps_counted)
    rows = [{'Name': n, 'TotalDurationNs': v[0], 'Calls': v[1], 'AverageNs': v[0] / v[1]} for n, v in agg.items()]
else:
    f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"Source: `{f.split('/')[-1]}`. Sum of kernel time = {tot/1e6:.1f} ms" + (f" = {tot/1e6/steps:.1f} ms/step over {steps:g} steps" if steps else ""))
print("\n| kernel | calls | total ms | avg us | % |" + (" ms/step |" if steps else "") + "\n|---|---|---|---|---|" + ("---|" if steps else ""))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); n = re.sub(r'\(.*$', '', n)
    return n[:90]
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:40]:
    t = float(r['TotalDurationNs'])
    print(f"| `{short(r['Name'])}` | {r['Calls']} | {t/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {100*t/tot:.2f} |" + (f" {t/1e6/steps:.2f} |" if steps else ""))
