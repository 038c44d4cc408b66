#!/bin/bash
# kernel trace of a few benchmark steps (one step in flight) + the kernel SEQUENCE of the last complete step (tools/trace_sequence.py)
# usage: gpurun -- bash tools/gpu_trace_step.sh TAG [extra bench args]
R=$PWD; T=${1:-x}; shift; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_$T -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-batch1 --in-flight 1 "$@" 2>&1 | tail -1) > $R/gpurun_out/trace_$T.log 2>&1
cd $R
F=$(find gpurun_out/trace_$T -name "*kernel_trace.csv" | head -1)
python tools/trace_sequence.py "$F" > gpurun_out/trace_${T}_sequence.txt 2>&1
S=$(find gpurun_out/trace_$T -name "*kernel_stats.csv" | head -1)
cp "$S" gpurun_out/trace_${T}_kernel_stats.csv
rm -rf gpurun_out/trace_$T
head -50 gpurun_out/trace_${T}_sequence.txt
