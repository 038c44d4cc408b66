#!/usr/bin/env python3
"""kernel sequence with gaps from a rocprofv3 --kernel-trace CSV: python tools/trace_gaps.py <kernel_trace.csv> [first] [count]"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
a = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
prev = None
for r in rows[a:a + n]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = re.sub(r'\(anonymous namespace\)::|^void ', '', r['Kernel_Name']); nm = re.sub(r'\(.*$', '', nm)[:40]
    print(f"{nm:40s} dur {(e - s) / 1e3:8.1f} us   gap {((s - prev) / 1e3 if prev else 0):8.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
    prev = e
