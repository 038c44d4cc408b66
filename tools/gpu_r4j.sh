#!/bin/bash
# pose step: characteristic-polynomial root finder vs the QR iteration of round 3 (same box), consensus kernel without spills;
# the graph tag wrap test; loops with the new pose step
TAG=${1:-r4j}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
V=$R/imp-release_amd/csrc/variants
(timeout 600 python -m pytest tests/test_gpu_pose.py tests/test_gpu_parity.py tests/test_gpu_ragged.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=300 -k "pose" 2>&1 | tail -8) > $O/${TAG}_tests.log 2>&1
for lib in "" $V/libimp_hip_poseqr.so; do
  echo "== library: ${lib:-default (root finder)}" >> $O/${TAG}_pose_time.log
  (IMP_HIP_LIB=$lib timeout 200 python tools/probe/pose_time.py 2>&1 | tail -3) >> $O/${TAG}_pose_time.log
done
cd /tmp && export TMPDIR=/tmp
for lib in "" $V/libimp_hip_poseqr.so; do
  n=$([ -z "$lib" ] && echo new || echo qr)
  rm -rf /tmp/prof_$n
  (cd $R && IMP_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o pose -- python tools/probe/pose_time.py > /dev/null 2>&1)
  f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats, library: $n" >> $O/${TAG}_pose_kernels.log
  [ -n "$f" ] && (head -1 $f; grep -i pose $f) >> $O/${TAG}_pose_kernels.log
done
cd $R
for lib in $V/libimp_hip_poseprof.so; do
  echo "== phases (x 10 ns), library: $(basename $lib)" >> $O/${TAG}_pose_phases.log
  (IMP_HIP_LIB=$lib timeout 200 python -c "
import sys; sys.path.insert(0, '.')
import torch; torch.zeros(1).cuda()
from imp_release_amd import pose
from oracle import pose_oracle as po
k0, k1, K, R, t, truth = po.synthetic_scene(1000, outliers=0.3, noise=0.3, seed=1)
pose.estimate_pose(k0, k1, K, K, 1.0); pose.estimate_pose(k0, k1, K, K, 1.0)
" 2>&1 | grep fivept | tail -4) >> $O/${TAG}_pose_phases.log
done
for lib in "" $V/libimp_hip_poseqr.so "" $V/libimp_hip_poseqr.so; do
  echo "== loops, library: ${lib:-default (root finder)}" >> $O/${TAG}_loops.log
  (IMP_HIP_LIB=$lib timeout 400 python tools/eval_synthetic.py --hard --model IMP --pairs 128 --lockstep 4 --workers 3 2>&1 | tail -1 | cut -c1-260) >> $O/${TAG}_loops.log
done
cat $O/${TAG}_tests.log $O/${TAG}_pose_time.log $O/${TAG}_pose_kernels.log $O/${TAG}_pose_phases.log $O/${TAG}_loops.log
