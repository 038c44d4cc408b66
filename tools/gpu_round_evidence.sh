#!/bin/bash
# round 6 evidence: the CU-holder rehearsal with its printed stalls, the lock-step difference probe, the conditioning sweep at N = 2048
R=$PWD; O=$R/gpurun_out/run; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_rehearsal.py -m gpu -q -s --no-header -p no:cacheprovider 2>&1 | grep -E "CU holder|soak:|passed|failed" | cut -c1-1500) > $O/rehearsal.log 2>&1
(timeout 600 python tools/probe/eimp_lockstep_diff.py 96 2>&1 | tail -3) > $O/lockstep_diff.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s --no-header -p no:cacheprovider -k "conditioning" 2>&1 | grep -v "^$" | tail -30 | cut -c1-400) > $O/conditioning.log 2>&1
cat $O/rehearsal.log $O/lockstep_diff.log; tail -12 $O/conditioning.log
