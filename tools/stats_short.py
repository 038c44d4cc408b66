#!/usr/bin/env python3
"""kernel_stats.csv of rocprofv3 --stats, shortened: python tools/stats_short.py <kernel_stats.csv> [rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    n = re.sub(r'\(anonymous namespace\)::|^void ', '', r['Name'])
    n = re.sub(r'\(.*$', '', n)[:60]
    print(f"{n:60s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  max {float(r['MaxNs']) / 1e3:9.1f}  {float(r['Percentage']):5.1f} %")
