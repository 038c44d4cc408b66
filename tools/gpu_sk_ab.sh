#!/bin/bash
# Sinkhorn A/B: per-iteration time + phase profile of the resident kernel for the default library and for variants
# usage: gpurun -- bash tools/gpu_sk_ab.sh TAG [variant tags under imp-release_amd/csrc/variants ...]
T=${1:-sk}; shift; mkdir -p gpurun_out; OUT=gpurun_out/sk_$T.log; : > $OUT
for rep in 1 2; do
  for v in "" "$@"; do
    if [ -n "$v" ]; then export IMP_HIP_LIB=$PWD/imp-release_amd/csrc/variants/libimp_hip_$v.so; else unset IMP_HIP_LIB; fi
    echo "=== [${v:-default}] pass $rep" >> $OUT
    IMP_OTR_PROF=1 timeout 200 python tools/probe/sk_prof.py 4 2048 1 1024 2>&1 | grep "IMP_OTR_PROF\|sinkhorn us" >> $OUT
  done
done
unset IMP_HIP_LIB
cat $OUT
