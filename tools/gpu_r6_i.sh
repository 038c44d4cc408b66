#!/bin/bash
R=$PWD; O=$R/gpurun_out/r6i; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_resident_ot.py tests/test_gpu_batch_invariance.py tests/test_gpu_hard_loops.py tests/test_gpu_ragged.py -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | cut -c1-400) > $O/tests.log 2>&1
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode 2>$O/bench.err | tail -1) > $O/bench.json
grep -E "passed|failed|^FAILED|^ERROR" $O/tests.log | tail; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6i/bench.json').read())
print({k:round(d[k],1) for k in d if (k.startswith('c5_') and isinstance(d[k],float)) or k in ('value','ragged_b4_pairs_per_s','eimp_n4096_ms_per_pair','adagmn_masked_n1024_b4_ms_per_call')}, d['roofline']['sinkhorn_iteration']['iteration_ms'])
PY
