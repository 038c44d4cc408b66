#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes (tools/gpu_pmc.sh) into per-kernel averages.

usage: python tools/pmc_summary.py TAG [out.csv]      reads gpurun_out/pmc_TAG_*/bench_counter_collection.csv
One row per (kernel, grid size): dispatch count, average duration and the per-dispatch average of every counter;
FETCH_SIZE / WRITE_SIZE are reported in KB as the profiler prints them (gfx950: double FETCH_SIZE, MI355X_MICROARCH.md).
"""
import csv, glob, re, sys
from collections import defaultdict

tag = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
for path in sorted(glob.glob(f'gpurun_out/pmc_{tag}_*/bench_counter_collection.csv')):
    seen = set()
    for r in csv.DictReader(open(path)):
        name = re.sub(r'^void |\(anonymous namespace\)::|\(.*$', '', r['Kernel_Name'])
        key = f"{name} grid={r['Grid_Size']}"
        a = acc[key][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
        d = (path, r['Dispatch_Id'])
        if d not in seen:
            seen.add(d)
            dur[key][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); dur[key][1] += 1
counters = sorted({c for k in acc for c in acc[k]})
rows = []
for k in acc:
    n = max(v[1] for v in acc[k].values())
    rows.append([k, n, dur[k][0] / max(dur[k][1], 1) / 1e3] + [acc[k][c][0] / acc[k][c][1] if acc[k][c][1] else '' for c in counters])
rows.sort(key=lambda r: -r[1] * r[2])
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
w = csv.writer(out)
w.writerow(['kernel', 'dispatches_per_pass', 'avg_us'] + counters)
for r in rows[:14]:
    w.writerow([r[0], r[1], '%.2f' % r[2]] + [('%.6g' % x if x != '' else '') for x in r[3:]])
