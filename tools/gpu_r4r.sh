#!/bin/bash
TAG=${1:-r4r}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
echo "== default" >> $O/${TAG}_sweep.log
(timeout 600 python tools/probe/c5_sweep.py 1920 4x3,4x4 2>&1 | grep "lockstep") >> $O/${TAG}_sweep.log
echo "== group_similar = window of 96 (the distinct scenes)" >> $O/${TAG}_sweep.log
(timeout 600 python tools/probe/c5_sweep.py 1920 4x3,4x4 similar 2>&1 | grep "lockstep") >> $O/${TAG}_sweep.log
cat $O/${TAG}_sweep.log
