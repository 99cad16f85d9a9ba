"""Kernel launches and time of ONE attack iteration: difference of two rocprofv3 kernel-stats files taken at 5 and 15 timed steps
(tools/lab/step_kernels.sh): python tools/lab/step_kernels.py gpurun_out/stepk/a_stats.csv gpurun_out/stepk/b_stats.csv"""
import csv, sys
a = {r['Name']: r for r in csv.DictReader(open(sys.argv[1]))}
b = {r['Name']: r for r in csv.DictReader(open(sys.argv[2]))}
rows = []
for n, rb in b.items():
    ra = a.get(n)
    dc = int(rb['Calls']) - (int(ra['Calls']) if ra else 0)
    dt = float(rb['TotalDurationNs']) - (float(ra['TotalDurationNs']) if ra else 0.0)
    if dc:
        rows.append((dt / 10 / 1e3, dc / 10, n))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f'{tot:9.1f} us of kernels per step, {sum(r[1] for r in rows):.1f} launches')
for us, c, n in rows:
    print(f'{us:9.1f} us {c:6.1f} x {us / c:8.1f} us  {n[:120]}')
