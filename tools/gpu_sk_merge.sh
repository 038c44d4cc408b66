#!/bin/bash
# resident Sinkhorn A/B of variant builds against the default: per-iteration time + phases, the resident / ragged tests, bench legs
# usage: gpurun -- bash tools/gpu_sk_merge.sh TAG variant [variant ...]     (variants: imp-release_amd/csrc/variants/libimp_hip_<variant>.so, tools/build_variant.sh)
T=${1:-x}; shift
O=gpurun_out/sk_$T.log; : > $O
bash tools/gpu_sk_ab.sh ${T}ab "$@" > /dev/null 2>&1; cat gpurun_out/sk_${T}ab.log >> $O
(timeout 600 python -m pytest tests/test_gpu_resident_ot.py tests/test_gpu_ragged.py -q -x --no-header -p no:cacheprovider 2>&1 | tail -4) >> $O
for v in "" "$@" ""; do
  if [ -n "$v" ]; then export IMP_HIP_LIB=$PWD/imp-release_amd/csrc/variants/libimp_hip_$v.so; else unset IMP_HIP_LIB; fi
  echo "=== [${v:-default}] bench" >> $O
  (timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-batch1 --no-f32-mode 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', round(d['value'],1), 'one', round((d.get('one_step_in_flight') or {}).get('value',0),1), 'sk it us', round(d['roofline']['sinkhorn_iteration']['iteration_ms']*1e3,2), d['config'].get('steps_in_flight_calibration'))") >> $O 2>&1
done
cat $O
