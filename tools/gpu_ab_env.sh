#!/bin/bash
# A/B of the benchmark under environment switches: usage  gpurun -- bash tools/gpu_ab_env.sh TAG "ENV1=a ENV2=b" "ENV1=c" ...   ("-" = no switch)
# every arm: python bench.py --no-cpu-baseline --no-batch1 --no-f32-mode (default steps), twice, alternating
T=${1:-ab}; shift; mkdir -p gpurun_out; OUT=gpurun_out/ab_$T.log; : > $OUT
for rep in 1 2; do
  for arm in "$@"; do
    [ "$arm" = "-" ] && envs="" || envs="$arm"
    line=$(env $envs timeout 300 python bench.py --no-cpu-baseline --no-batch1 --no-f32-mode 2>/dev/null | tail -1)
    echo "$line" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('arm [%s] pass $rep: value %.1f  one_in_flight %.1f  in_flight %s  attn %.1f us  sclk %s' % ('$arm', d['value'], d['one_step_in_flight']['value'], d['config']['steps_in_flight_per_gpu'], d['roofline']['launch_ms']*1e3, d['roofline'].get('sclk_mhz_observed')))" >> $OUT 2>&1
  done
done
cat $OUT
