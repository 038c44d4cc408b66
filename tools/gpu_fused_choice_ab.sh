#!/bin/bash
# when does a stream count as alone (and take the fused layer launch)?  after seven kernel choices in a row (default) against after 8 ms (option fused_choice = 1), same box, interleaved
R=$PWD; O=$R/gpurun_out/fused_choice; mkdir -p $O; : > $O/ab.log
for i in 1 2 3; do
  for E in "-" "fused_choice=1"; do
    if [ "$E" = "-" ]; then L=$(timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-batch1 --no-f32-mode 2>/dev/null | tail -1); else L=$(IMP_OPTIONS=$E timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-batch1 --no-f32-mode 2>/dev/null | tail -1); fi
    python - "$E" "$L" >> $O/ab.log <<'PY'
import json,sys
d=json.loads(sys.argv[2])
c=d['config']
print('%-16s pairs/s %7.1f  ms/step %.3f  in-flight %d  calibration %s  one-in-flight %s  attn %.2f us  sclk %.0f' % (sys.argv[1], d['value'], d['ms_per_step'], c['steps_in_flight_per_gpu'], {k[12:]:round(v,1) for k,v in (c['steps_in_flight_calibration'] or {}).items() if k.startswith('steps_per_s')}, round(d.get('one_step_in_flight',{}).get('value',0),1) if isinstance(d.get('one_step_in_flight'),dict) else d.get('one_step_in_flight'), d['roofline']['launch_ms']*1e3, d['roofline']['sclk_mhz_observed']))
PY
  done
done
cat $O/ab.log
