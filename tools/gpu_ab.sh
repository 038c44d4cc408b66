#!/bin/bash
# same-box A/B of bench.py under environment switches:  gpurun -- bash tools/gpu_ab.sh TAG "ENV1=.. ENV2=.." "ENV..." ...   ("-" = no switch)
TAG=$1; shift
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for rep in $(seq 1 ${REPS:-2}); do
  for E in "$@"; do
    [ "$E" = "-" ] && E=""
    (env $E timeout 300 python bench.py --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline --no-batch1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-40s pairs/s %.1f  ms/step %.3f  one-in-flight %.1f  attn %.1f us  sk-iter %.2f us' % ('$E' or '(default)', d['value'], d['ms_per_step'], d['one_step_in_flight']['value'], r['launch_ms']*1e3, r['sinkhorn_iteration']['iteration_ms']*1e3))
") >> $O/ab_$TAG.log 2>&1
  done
done
cat $O/ab_$TAG.log
