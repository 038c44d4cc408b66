#!/usr/bin/env python3
"""Kernel sequence of one benchmark step from a rocprofv3 --kernel-trace CSV: the launches between the last two `mutual_kernel`s, in start
order, runs of the same kernel compressed, with the gap to the previous kernel's end.   python tools/trace_sequence.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n)[:70]


idx = [i for i, r in enumerate(rows) if 'mutual_kernel' in r['Kernel_Name']]
if len(idx) < 2:
    sys.exit('fewer than two mutual_kernel launches in the trace')
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0 = int(rows[a - 1]['End_Timestamp'])
print(f'step = {len(step)} launches, {(int(step[-1]["End_Timestamp"]) - t0) / 1e3:.1f} us from the end of the previous step')
prev_end, out = t0, []
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append((short(r['Kernel_Name']), (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Stream_Id', '?')))
    prev_end = max(prev_end, e)
busy = sum(o[1] for o in out)
print(f'sum of kernel durations {busy:.1f} us, sum of positive gaps {sum(max(0, o[2]) for o in out):.1f} us')
print(f'{"kernel":70s} {"us":>8s} {"gap us":>8s} {"grid":>8s} stream')
for o in out:
    print(f'{o[0]:70s} {o[1]:8.1f} {o[2]:8.1f} {o[3]:>8s} {o[4]}')
agg = {}
for o in out:
    k = agg.setdefault(o[0], [0, 0.0])
    k[0] += 1; k[1] += o[1]
print('\nper kernel in this step:')
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'  {k:70s} {n:4d} x  {t:9.1f} us')
