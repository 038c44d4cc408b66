#!/bin/bash
# configs[4] loops: where did round 6 lose against round 5?  the same bench command on both trees (_r5tree = git archive d6c23dd), plain and under rocprofv3
R=$PWD; O=$R/gpurun_out/r6c5; mkdir -p $O
for T in r6 r5; do
  D=$R; [ $T = r5 ] && D=$R/_r5tree
  (cd $D && timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mode 2>$O/err_$T.log | tail -1) > $O/line_$T.json
done
cd /tmp && export TMPDIR=/tmp
for T in r6 r5; do
  D=$R; [ $T = r5 ] && D=$R/_r5tree
  (cd $D && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mode --quick-c5 2>$O/prof_err_$T.log | tail -1) > $O/prof_line_$T.json
  S=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" $O/kernel_stats_$T.csv
  rm -rf $O/prof_$T
done
cd $R
python - <<'PY'
import json,csv,os
for T in ('r6','r5'):
    try:
        d=json.loads(open(f'gpurun_out/r6c5/line_{T}.json').read())
        print(T,{k:round(d[k],1) for k in d if k.startswith('c5_') and isinstance(d[k],float)})
    except Exception as e: print(T,'no line',e)
    f=f'gpurun_out/r6c5/kernel_stats_{T}.csv'
    if os.path.exists(f):
        rows=list(csv.DictReader(open(f)))
        tot=sum(float(r['TotalDurationNs']) for r in rows)
        print(T,'total kernel ms',tot/1e6)
        for r in rows[:30]: print('   ',r['Name'][:90],r['Calls'],round(float(r['TotalDurationNs'])/1e6,1),round(float(r['AverageNs'])/1e3,1))
    else:
        print(open(f'gpurun_out/r6c5/prof_err_{T}.log').read()[-1500:])
PY
