#!/bin/bash
# PMC pass (own run: --pmc with kernel-trace only).  usage: gpurun -- bash tools/gpu_pmc.sh TAG "COUNTERS..." ["COUNTERS2..."]
TAG=$1; shift
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "(SQ|TCC|TCP|TA|GRBM)_[A-Z0-9_]+" | sort -u > $R/gpurun_out/counters_avail.txt
i=0
for C in "$@"; do
  i=$((i+1))
  (timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 --in-flight 1 2>&1 | tail -2) > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
done
cd $R
wc -l gpurun_out/counters_avail.txt; ls gpurun_out/pmc_${TAG}_*/ | head
