#!/bin/bash
# attention prologue / epilogue tweaks: tests, launch time (bench roofline leg), bench
TAG=${1:-r4g}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=300 -k "attention or kv or golden or full_size or bench_batch or ragged" 2>&1 | tail -8) > $O/${TAG}_tests.log 2>&1
for r in 1 2; do (timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-batch1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pairs/s %.1f  one-in-flight %.1f  attn %.2f us frac %.4f' % (d['value'], d['one_step_in_flight']['value'], d['roofline']['launch_ms']*1e3, d['roofline']['frac']))") >> $O/${TAG}_bench.log 2>&1; done
cat $O/${TAG}_tests.log $O/${TAG}_bench.log
