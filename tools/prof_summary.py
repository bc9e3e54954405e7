"""Summarise a rocprofv3 `--kernel-trace --stats --output-format csv` run as a markdown table.
usage: python tools/prof_summary.py <dir-with-*_kernel_stats.csv> [steps]"""
import csv, glob, re, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
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
