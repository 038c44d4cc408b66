#!/bin/bash
# sweep of the count rule: a stream is alone after more than K kernel choices in a row (default 6)
R=$PWD; O=$R/gpurun_out/fused_choice; mkdir -p $O; : > $O/sweep.log
for i in 1 2; do
  for K in 6 2 4 9 13 18 30; do
    L=$(IMP_OPTIONS=fused_alone_after=$K timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-batch1 --no-f32-mode 2>/dev/null | tail -1)
    python - "$K" "$L" >> $O/sweep.log <<'PY'
import json,sys
d=json.loads(sys.argv[2]); c=d['config']
print('alone after > %2s  pairs/s %7.1f  in-flight %d  calibration %s' % (sys.argv[1], d['value'], c['steps_in_flight_per_gpu'], {k[12:]:round(v,1) for k,v in (c['steps_in_flight_calibration'] or {}).items() if k.startswith('steps_per_s')}))
PY
  done
done
cat $O/sweep.log
