#!/bin/bash
# lanes per five-point sample: 64 (default) vs 32 / 16 - pose call latency, then the loops (chip load)
TAG=${1:-r4v}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
V=$R/imp-release_amd/csrc/variants
for lib in "" $V/libimp_hip_fpl32.so $V/libimp_hip_fpl16.so; do
  echo "== library: ${lib:-default (64 lanes per sample)}" >> $O/${TAG}.log
  (IMP_HIP_LIB=$lib timeout 300 python -m pytest tests/test_gpu_pose.py -m gpu -q --no-header -p no:cacheprovider -x -k "twin" 2>&1 | tail -1) >> $O/${TAG}.log
  (IMP_HIP_LIB=$lib timeout 200 python tools/probe/pose_time.py 2>&1 | grep "n=" | cut -c1-110) >> $O/${TAG}.log
  (IMP_HIP_LIB=$lib timeout 600 python tools/probe/c5_sweep.py 1600 4x4 2>&1 | grep "lockstep") >> $O/${TAG}.log
done
cat $O/${TAG}.log
