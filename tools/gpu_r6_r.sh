#!/bin/bash
# hunt for the residual time-out of a waiting launch: the loops with four groups in flight, again and again; post-mortems of whatever is met
R=$PWD; O=$R/gpurun_out/r6r; mkdir -p $O; : > $O/hunt.log
for i in 1 2 3 4 5 6; do
  for L in eimp imp; do (timeout 400 python tools/probe/c5_probe.py $L 4 4000 4 2>&1 | grep "^c5\|^replica") >> $O/hunt.log; done
done
cat $O/hunt.log | cut -c1-1200
