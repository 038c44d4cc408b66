#!/bin/bash
# (needs the round-5 tree beside this one: mkdir _r5tree && git archive d6c23dd | tar -x -C _r5tree && make -C _r5tree/imp-release_amd/csrc)
# c5 loops: r6 tree vs r5 tree, groups in flight 1 / 2 / 4, lock step 4 and single pairs
R=$PWD; O=$R/gpurun_out/run; mkdir -p $O; : > $O/c5.log
for T in r6 r5; do
  D=$R; [ $T = r5 ] && D=$R/_r5tree
  for L in imp eimp; do
    for W in 1 2 4; do
      (cd $D && echo -n "$T " && timeout 300 python $R/tools/probe/c5_probe.py $L $W 600 4 2>&1 | grep "^c5") >> $O/c5.log
    done
    (cd $D && echo -n "$T " && timeout 300 python $R/tools/probe/c5_probe.py $L 1 200 1 2>&1 | grep "^c5") >> $O/c5.log
  done
done
cat $O/c5.log
