#!/bin/bash
# same-box A/B of bench.py under environment switches, lean legs only:  gpurun -- bash tools/gpu_ab_env.sh TAG "ENV1=.. ENV2=.." "ENV..." ...   ("-" = no switch; REPS, STEPS)
TAG=$1; shift
R=$PWD; O=$R/gpurun_out; mkdir -p $O; L=$O/envab_$TAG.log; : > $L
for rep in $(seq 1 ${REPS:-3}); do
  for E in "$@"; do
    [ "$E" = "-" ] && E=""
    (env $E timeout 200 python bench.py --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline --no-batch1 --no-f32-mode 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
g=d.get('layer_gemm_us') or {}
print('%-60s pairs/s %.1f  ms/step %.3f  in-flight %d  one-in-flight %.1f  attn %.2f us  frac %.3f  sclk %.0f  sk %.2f us  fused-layer %.1f us  two-launch %.1f us' % ('$E' or '(default)', d['value'], d['ms_per_step'], d['config']['steps_in_flight_per_gpu'], (d['one_step_in_flight'] or {}).get('value') or 0, r['launch_ms']*1e3, r['frac'], r.get('sclk_mhz_observed') or 0, r['sinkhorn_iteration']['iteration_ms']*1e3, g.get('fused_launch') or 0, g.get('two_launches') or 0))
") >> $L 2>&1
  done
done
cat $L
