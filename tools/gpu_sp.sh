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
head -20 "$f" | cut -c1-200
