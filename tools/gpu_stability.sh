#!/bin/bash
# stability of the last tree: the GPU suite twice, the opt-in soak module, the driver's bench command twice
R=$PWD; O=$R/gpurun_out/run; mkdir -p $O
for i in 1 2; do (timeout 1500 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | cut -c1-300 | tail -60) > $O/suite_$i.log 2>&1; grep -E "passed|failed" $O/suite_$i.log | tail -1; done
(IMP_SOAK=2 timeout 1500 python -m pytest tests/test_gpu_soak.py -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | cut -c1-300 | tail -15) > $O/soak.log 2>&1; grep -E "passed|failed" $O/soak.log | tail -1
for i in 1 2; do (timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_$i.err | tail -1) > $O/bench_$i.json; python - $O/bench_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
print({k:(round(d[k],1) if isinstance(d[k],float) else d[k]) for k in ('value','ragged_b4_pairs_per_s','c5_imp_pairs_per_s','c5_imp_spread','c5_eimp_pairs_per_s','c5_eimp_spread','c5_imp_voided_launches','c5_eimp_voided_launches','voided_launches_on_the_headline_replicas','c2_latency_ms','eimp_n4096_ms_per_pair')}, round(d['roofline']['frac'],3), d['config']['steps_in_flight_per_gpu'])
PY
done
