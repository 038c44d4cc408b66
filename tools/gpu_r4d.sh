#!/bin/bash
# ragged batches: tests + the whole GPU suite + bench line
TAG=${1:-r4d}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_ragged.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=200 2>&1 | tail -40) > $O/${TAG}_ragged.log 2>&1
(timeout 600 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=300 --deselect tests/test_gpu_ragged.py 2>&1 | tail -25) > $O/${TAG}_all.log 2>&1
(timeout 400 python bench.py --no-cpu-baseline --quick-c5 2>&1 | tail -1) > $O/${TAG}_bench.json 2>&1
cat $O/${TAG}_ragged.log; tail -12 $O/${TAG}_all.log; python -c "
import json; d=json.load(open('$O/${TAG}_bench.json')); print(d['value'], d['config']['steps_in_flight_calibration'], d['one_step_in_flight']['value'], d['layer_gemm_us'], d.get('ragged_b4_pairs_per_s'), d.get('ragged_b4_note'), d.get('c2_latency_ms'), d.get('eimp_n4096_ms_per_pair'))"
