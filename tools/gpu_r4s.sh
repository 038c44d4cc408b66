#!/bin/bash
TAG=${1:-r4s}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 900 python tools/probe/c5_sweep.py 1920 4x4,4x5,4x6,3x6,2x8 2>&1 | grep "lockstep") >> $O/${TAG}_sweep.log
cat $O/${TAG}_sweep.log
