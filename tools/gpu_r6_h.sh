#!/bin/bash
# after the prune: the GPU suite and the bench line must be what they were
R=$PWD; O=$R/gpurun_out/r6h; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | cut -c1-400) > $O/suite.log 2>&1
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | tail -1) > $O/bench_driver_command.json
grep -E "passed|failed|^FAILED|^ERROR" $O/suite.log | tail -20; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6h/bench_driver_command.json').read())
print({k:d[k] for k in ('value','ms_per_step','c5_imp_pairs_per_s','c5_eimp_pairs_per_s','c2_latency_ms')}, d['roofline']['frac'], d['roofline']['launch_ms'])
PY
