#!/bin/bash
R=$PWD; O=$R/gpurun_out/r6k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in imp; do
  (cd $R && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r6n_$L -o c5 -- python $R/tools/probe/c5_probe.py $L 1 200 4 2>/dev/null | grep "^c5") > $O/line_r6n_$L.txt
  S=$(find $O/prof_r6n_$L -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" $O/kernel_stats_r6n_$L.csv
  T=$(find $O/prof_r6n_$L -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && python3 - "$T" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
# attention launches: duration by grid size
by=collections.defaultdict(list)
for r in rows:
    if 'attn_f16x3_pp' in r['Kernel_Name']:
        by[int(r['Grid_Size_X'])//512 if 'Grid_Size_X' in r else int(r['Grid_Size'])//512].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k in sorted(by): print('attn grid %4d wgs: %4d launches avg %.1f us'%(k,len(by[k]),sum(by[k])/len(by[k])))
PY
  rm -rf $O/prof_r6n_$L
done
cd $R; cat $O/line_r6n_imp.txt; head -8 $O/kernel_stats_r6n_imp.csv | cut -c1-140
