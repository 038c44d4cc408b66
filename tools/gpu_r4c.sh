#!/bin/bash
# fused layer kernel v2: tests, phase cycles, bench (auto in-flight) + A/B
TAG=${1:-r4c}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=200 -k "fused or in_flight" 2>&1 | tail -12) > $O/${TAG}_tests.log 2>&1
(IMP_WF_PROF=1 IMP_HIP_LIB=$R/imp-release_amd/csrc/variants/libimp_hip_wfprof.so timeout 200 python tools/probe/fused_time.py 2>&1 | grep -v amdgpu.ids | tail -9) > $O/${TAG}_phases.log 2>&1
(timeout 300 python bench.py --no-cpu-baseline --no-batch1 2>&1 | tail -1) > $O/${TAG}_bench.json 2>&1
REPS=1 STEPS=40 bash tools/gpu_ab.sh $TAG "-" "IMP_WF_FUSED=0" "IMP_WF_FUSED=2" > /dev/null 2>&1
for K in 1 2 3; do (timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-batch1 --in-flight $K 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in-flight $K: %.1f pairs/s' % d['value'], d.get('layer_gemm_us'))") >> $O/${TAG}_inflight.log 2>&1; done
cat $O/${TAG}_tests.log $O/${TAG}_phases.log; cut -c1-700 $O/${TAG}_bench.json; echo; python -c "
import json; d=json.load(open('$O/${TAG}_bench.json')); print(d['value'], d['config']['steps_in_flight_calibration'], d['one_step_in_flight'], d['layer_gemm_us'])"; cat $O/ab_$TAG.log $O/${TAG}_inflight.log
