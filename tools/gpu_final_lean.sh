#!/bin/bash
# lean end-of-round check (one gpurun call, ~8 min): GPU suite, smoke, default bench line, the driver's bench setting, per-kernel stats
R=$PWD; T=${1:-x}; O=$R/gpurun_out; mkdir -p $O
(timeout 600 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | tail -6) > $O/final_${T}_pytest.log 2>&1
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $O/final_${T}_smoke.log 2>&1
(timeout 900 python bench.py 2>&1 | tail -1) > $O/final_${T}_bench.json 2>&1
(timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-batch1 2>&1 | tail -1) > $O/final_${T}_bench_driverlike.json 2>&1
cd /tmp && export TMPDIR=/tmp
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_${T}_prof1 -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 --in-flight 1 2>&1 | tail -1) > $O/final_${T}_rocprof1.log 2>&1
cd $R
(timeout 200 python tools/gpu_configs.py 2>&1 | tail -7) > $O/final_${T}_configs.log 2>&1
tail -4 $O/final_${T}_pytest.log; tail -2 $O/final_${T}_smoke.log; cut -c1-3000 $O/final_${T}_bench.json; echo; cut -c1-600 $O/final_${T}_bench_driverlike.json; echo; head -9 $O/final_${T}_prof1/bench_kernel_stats.csv | cut -c1-150; cat $O/final_${T}_configs.log
