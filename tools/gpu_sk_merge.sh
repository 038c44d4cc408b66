#!/bin/bash
# resident Sinkhorn A/B of a variant build against the default (round 5: tools/build_variant.sh merge -DOTR_MERGE=1): per-iteration time + phases, resident tests, bench legs
# usage: gpurun -- bash tools/gpu_sk_merge.sh   (expects imp-release_amd/csrc/variants/libimp_hip_nomerge.so; round 5 ran it with the roles swapped: default = merged)
O=gpurun_out/sk_merge.log; : > $O
bash tools/gpu_sk_ab.sh mergeab nomerge > /dev/null 2>&1; cat gpurun_out/sk_mergeab.log >> $O
(timeout 600 python -m pytest tests/test_gpu_resident_ot.py tests/test_gpu_ragged.py -q -x --no-header -p no:cacheprovider 2>&1 | tail -4) >> $O
for v in "" nomerge ""; do
  if [ -n "$v" ]; then export IMP_HIP_LIB=$PWD/imp-release_amd/csrc/variants/libimp_hip_$v.so; else unset IMP_HIP_LIB; fi
  echo "=== [${v:-default}] bench" >> $O
  (timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-batch1 --no-f32-mode 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', round(d['value'],1), 'one', round((d.get('one_step_in_flight') or {}).get('value',0),1), 'sk it us', round(d['roofline']['sinkhorn_iteration']['iteration_ms']*1e3,2), d['config'].get('steps_in_flight_calibration'))") >> $O 2>&1
done
cat $O
