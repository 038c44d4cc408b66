#!/bin/bash
# rocprofv3 kernel-trace stats of (a) the bench workload, one step in flight, (b) BASELINE configs[1] (N=1024, batch 1)
# usage: tools/gpu_prof.sh <tag>
R=$PWD; T=${1:-x}; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_bench -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 --in-flight 1 2>&1 | tail -1) > $R/gpurun_out/prof_${T}_bench.log 2>&1
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_c2 -o c2 -- python $R/tools/probe/c2_run.py 1024 1 2>&1 | tail -1) > $R/gpurun_out/prof_${T}_c2.log 2>&1
cd $R
for f in gpurun_out/prof_${T}_bench/bench_kernel_stats.csv gpurun_out/prof_${T}_c2/c2_kernel_stats.csv; do echo == $f; head -16 $f | cut -c1-170; done
