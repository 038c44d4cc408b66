#!/bin/bash
# quick GPU iteration: parity tests, short bench, rocprof kernel stats  (usage: gpurun -- bash tools/gpu_quick.sh TAG)
TAG=${1:-quick}
R=$PWD
mkdir -p gpurun_out
(timeout 300 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | tail -30) > gpurun_out/pytest_$TAG.log 2>&1
(timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/bench_$TAG.log 2>&1
cd /tmp && export TMPDIR=/tmp
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --in-flight 1 2>&1 | tail -1) > $R/gpurun_out/rocprof_$TAG.log 2>&1
cd $R
tail -4 gpurun_out/pytest_$TAG.log
python - <<PY
import json
for l in open('gpurun_out/bench_$TAG.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('pairs/s %.1f  ms/step %.2f  attn %.1f TF (%.3f ms)  whole %.1f TF  sinkhorn iter %.1f us' % (d['value'], d['ms_per_step'], r['achieved'], r['launch_ms'], r['whole_path_tflops'], r['sinkhorn_iteration']['iteration_ms']*1e3))
PY
head -7 gpurun_out/prof_$TAG/bench_kernel_stats.csv | cut -c1-160
