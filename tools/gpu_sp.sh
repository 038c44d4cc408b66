#!/bin/bash
# SuperPoint front-end: tests + timing + per-kernel profile (gpurun -- 'bash tools/gpu_sp.sh')
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/sp
timeout 280 python -m pytest tests/test_gpu_superpoint.py -q 2>&1 | tail -15
python tools/probe/sp_time.py 480 640 1024 50 | tee gpurun_out/sp/time.log
python tools/probe/sp_time.py 480 640 1024 50 4 | tee -a gpurun_out/sp/time.log
python tools/probe/sp_time.py 240 320 -1 50 | tee -a gpurun_out/sp/time.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/spprof -o sp -- python $R/tools/probe/sp_time.py 480 640 1024 30 > /tmp/spprof.log 2>&1
f=$(find /tmp/spprof -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/sp/sp_480x640_kernel_stats.csv 2>/dev/null
head -24 "$f" | cut -c1-200
t=$(find /tmp/spprof -name '*kernel_trace.csv' | head -1)
python - "$t" <<'PY' | tee $R/gpurun_out/sp/sp_480x640_per_layer.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print('columns:', list(rows[0].keys()) if rows else 'NO ROWS')
agg = collections.OrderedDict()
for r in rows:
    name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    if 'sp_' not in name:
        continue
    key = (name, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', '?'))
    agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('per dispatch (kernel, grid threads): n, median us')
for (name, grid), v in agg.items():
    v = sorted(v)
    print(f'  {name:40s} grid {grid:>9s}  n={len(v):3d}  median {v[len(v) // 2]:8.2f} us')
PY
