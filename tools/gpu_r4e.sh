#!/bin/bash
# lock-step loops + harder configs[4]
TAG=${1:-r4e}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 400 python -m pytest tests/test_gpu_ragged.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=300 -k "lockstep" -s 2>&1 | tail -40) > $O/${TAG}_lock.log 2>&1
(timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1) > $O/${TAG}_bench.json 2>&1
cat $O/${TAG}_lock.log; python -c "
import json; d=json.load(open('$O/${TAG}_bench.json'))
for k in ('value','c5_imp_pairs_per_s','c5_imp_second_run_pairs_per_s','c5_imp_single_pairs_3_in_flight_pairs_per_s','c5_imp_report','c5_eimp_pairs_per_s','c5_eimp_second_run_pairs_per_s','c5_eimp_report','ragged_b4_pairs_per_s'): print(k, d.get(k))"
