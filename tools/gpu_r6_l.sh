#!/bin/bash
R=$PWD; O=$R/gpurun_out/r6l; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider -x 2>&1 | cut -c1-400) > $O/suite.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $O/suite.log | tail -5
: > $O/c5.log
for L in imp eimp; do for W in 1 4; do (timeout 300 python tools/probe/c5_probe.py $L $W 600 4 2>&1 | grep "^c5") >> $O/c5.log; done; (timeout 300 python tools/probe/c5_probe.py $L 3 300 1 2>&1 | grep "^c5") >> $O/c5.log; done
cat $O/c5.log
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode --quick-c5 2>$O/bench.err | tail -1) > $O/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6l/bench.json').read())
print({k:round(d[k],2) for k in ('value','ragged_b4_pairs_per_s','eimp_n4096_ms_per_pair','adagmn_masked_n1024_b4_ms_per_call','c2_latency_ms','c2_latency_graph_ms')}, d['roofline']['launch_ms'], d['roofline']['frac'])
PY
