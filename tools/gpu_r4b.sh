#!/bin/bash
# fused layer kernel: phase cycles (WF_PROFILE variant) + kernel trace of the bench workload, one step in flight, fused on / off
TAG=${1:-r4b}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(IMP_WF_PROF=1 IMP_HIP_LIB=$R/imp-release_amd/csrc/variants/libimp_hip_wfprof.so timeout 200 python tools/probe/fused_time.py 2>&1 | grep -v amdgpu.ids) > $O/${TAG}_phases.log 2>&1
cd /tmp && export TMPDIR=/tmp
for F in 1 0; do
(IMP_WF_FUSED=$F timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_f$F -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batch1 --in-flight 1 2>&1 | tail -1 | cut -c1-300) > $O/${TAG}_rocprof_f$F.log 2>&1
done
cd $R
cat $O/${TAG}_phases.log
for F in 1 0; do echo "--- IMP_WF_FUSED=$F"; head -8 $O/${TAG}_prof_f$F/bench_kernel_stats.csv | cut -c1-170; done
