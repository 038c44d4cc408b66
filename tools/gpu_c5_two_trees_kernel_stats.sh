#!/bin/bash
# (needs the round-5 tree beside this one: mkdir _r5tree && git archive d6c23dd | tar -x -C _r5tree && make -C _r5tree/imp-release_amd/csrc)
# c5 loops, one group in flight: kernel stats on both trees
R=$PWD; O=$R/gpurun_out/run; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for T in r6 r5; do
  D=$R; [ $T = r5 ] && D=$R/_r5tree
  for L in imp eimp; do
    (cd $D && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}_$L -o c5 -- python $R/tools/probe/c5_probe.py $L 1 200 4 2>/dev/null | grep "^c5") > $O/line_${T}_$L.txt
    S=$(find $O/prof_${T}_$L -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" $O/kernel_stats_${T}_$L.csv
    rm -rf $O/prof_${T}_$L
  done
done
cd $R
python - <<'PY'
import csv,re
def load(f):
    d={}
    for r in csv.DictReader(open(f)):
        n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'\(.*$','',n).replace('void ','')
        n=n.replace('<64, false, false>','<64, false>')
        d[n]=(int(r['Calls']),float(r['TotalDurationNs'])/1e6)
    return d
for L in ('imp','eimp'):
    a=load(f'gpurun_out/run/kernel_stats_r6_{L}.csv'); b=load(f'gpurun_out/run/kernel_stats_r5_{L}.csv')
    print(L, open(f'gpurun_out/run/line_r6_{L}.txt').read().strip(), '|', open(f'gpurun_out/run/line_r5_{L}.txt').read().strip())
    print(L,'total ms r6 %.1f r5 %.1f'%(sum(v[1] for v in a.values()),sum(v[1] for v in b.values())))
    keys=sorted(set(a)|set(b),key=lambda k:-abs(a.get(k,(0,0))[1]-b.get(k,(0,0))[1]))
    for k in keys[:18]:
        x=a.get(k,(0,0)); y=b.get(k,(0,0))
        print('   %-60s r6 %5d calls %8.1f ms | r5 %5d calls %8.1f ms | diff %+7.1f'%(k[:60],x[0],x[1],y[0],y[1],x[1]-y[1]))
PY
