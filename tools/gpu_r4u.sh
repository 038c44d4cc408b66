#!/bin/bash
# where the GPU time of the configs[4] loops goes: kernel stats of 600 evaluations each (IMP, EIMP; groups of 4 x 4 in flight)
TAG=${1:-r4u}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o loops -- python $R/tools/probe/c5_sweep.py 600 4x4 2>&1 | grep "lockstep") > $O/${TAG}_rates.log 2>&1
cd $R
python - <<PY
import csv
rows = list(csv.DictReader(open('$O/${TAG}_prof/loops_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time %.1f ms over %d kernels' % (tot / 1e6, len(rows)))
for r in rows[:22]:
    print('%-72s calls %6s avg %8.1f us  %5.1f %%' % (r['Name'].replace('(anonymous namespace)::', '')[:72], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
cat $O/${TAG}_rates.log
