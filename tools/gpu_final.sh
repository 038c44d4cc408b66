#!/bin/bash
# driver-like final check: GPU tests, smoke, default bench (with CPU baseline), rocprof kernel stats
R=$PWD; mkdir -p gpurun_out
(timeout 400 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/final_pytest.log 2>&1
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > gpurun_out/final_smoke.log 2>&1
(timeout 600 python bench.py 2>&1 | tail -1) > gpurun_out/final_bench.json 2>&1
cd /tmp && export TMPDIR=/tmp
# per-kernel evidence is taken with ONE step in flight: overlapped kernels share the GPU, their durations are then not those of a full-chip launch
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --in-flight 1 2>&1 | tail -1) > $R/gpurun_out/final_rocprof.log 2>&1
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final3 -o bench -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > $R/gpurun_out/final_rocprof3.log 2>&1
cd $R
tail -3 gpurun_out/final_pytest.log; cat gpurun_out/final_smoke.log | tail -2; cut -c1-1500 gpurun_out/final_bench.json; echo; head -8 gpurun_out/prof_final/bench_kernel_stats.csv | cut -c1-150
