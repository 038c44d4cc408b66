#!/bin/bash
# fused v3 (two-stage overlapped exchange) vs v2: bit-identity tests, timing A/B, phase cycles; lock-step loops after the host fixes
TAG=${1:-r4f}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_gpu_ragged.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=200 -k "fused or in_flight or ragged or lockstep" 2>&1 | tail -12) > $O/${TAG}_tests.log 2>&1
for V in v3 v2 v3 v2; do
  if [ $V = v2 ]; then export IMP_HIP_LIB=$R/imp-release_amd/csrc/variants/libimp_hip_fusedv2.so; else unset IMP_HIP_LIB; fi
  (echo "== $V"; timeout 200 python tools/probe/fused_time.py 2>&1 | grep -v amdgpu.ids | tail -3) >> $O/${TAG}_time.log 2>&1
  (timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-batch1 --in-flight 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V one in flight: %.1f pairs/s' % d['value'], d['layer_gemm_us']['fused_launch'])") >> $O/${TAG}_time.log 2>&1
done
unset IMP_HIP_LIB
(IMP_WF_PROF=1 IMP_HIP_LIB=$R/imp-release_amd/csrc/variants/libimp_hip_wfprof.so timeout 200 python tools/probe/fused_time.py 2>&1 | grep -v amdgpu.ids | grep fused | tail -4) > $O/${TAG}_phases.log 2>&1
(timeout 300 python tools/probe/lockstep_profile.py 2>&1 | grep -v amdgpu.ids | grep "lockstep") > $O/${TAG}_lockstep.log 2>&1
cat $O/${TAG}_tests.log $O/${TAG}_time.log $O/${TAG}_phases.log $O/${TAG}_lockstep.log
