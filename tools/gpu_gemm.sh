#!/bin/bash
# layer-GEMM iteration: parity of the GEMM paths, per-kernel timing probe (default build and variants), short bench A/B
#   usage: gpurun -- bash tools/gpu_gemm.sh TAG [variant ...]
TAG=${1:-g}; shift
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | tail -15) > $O/pytest_$TAG.log 2>&1
(timeout 120 python tools/probe/gemm_wf_time.py 2>&1 | grep -v amdgpu.ids) > $O/gemmtime_$TAG.log 2>&1
(timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-batch1 2>&1 | tail -1) > $O/bench_$TAG.log 2>&1
for V in "$@"; do
  L=$R/imp-release_amd/csrc/variants/libimp_hip_$V.so
  (IMP_HIP_LIB=$L timeout 120 python tools/probe/gemm_wf_time.py 4 2048 2>&1 | grep -v amdgpu.ids) > $O/gemmtime_${TAG}_$V.log 2>&1
  (IMP_HIP_LIB=$L timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-batch1 2>&1 | tail -1) > $O/bench_${TAG}_$V.log 2>&1
done
(IMP_WF_CHAIN=0 timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-batch1 2>&1 | tail -1) > $O/bench_${TAG}_nochain.log 2>&1
(IMP_GEMM_WF=2 timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-batch1 2>&1 | tail -1) > $O/bench_${TAG}_wfall.log 2>&1
tail -5 $O/pytest_$TAG.log; cat $O/gemmtime_$TAG*.log
python - <<PY
import json, glob
for f in sorted(glob.glob('$O/bench_${TAG}*.log')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], 'pairs/s %.1f  ms/step %.3f  one-in-flight %.1f' % (d['value'], d['ms_per_step'], d.get('one_step_in_flight', {}).get('value', 0)))
PY
