#!/bin/bash
# (needs the round-5 tree beside this one: mkdir _r5tree && git archive d6c23dd | tar -x -C _r5tree && make -C _r5tree/imp-release_amd/csrc)
# resident Sinkhorn: phase cycles per iteration, round 6 tree against the round-5 tree on one box
R=$PWD; O=$R/gpurun_out/run; mkdir -p $O; : > $O/sk.log
for i in 1 2; do
  (cd $R && echo -n "r6 " && IMP_OPTIONS=probe_prof=1 timeout 200 python tools/probe/sk_prof.py 4 2048 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo) >> $O/sk.log
  (cd $R/_r5tree && echo -n "r5 " && IMP_OTR_PROF=1 timeout 200 python tools/probe/sk_prof.py 4 2048 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo) >> $O/sk.log
done
cat $O/sk.log | cut -c1-400
