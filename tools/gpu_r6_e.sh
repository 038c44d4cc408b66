#!/bin/bash
# round 6, session 3: where the tree stands - full GPU suite (unfiltered log), lock-step diff, bench line
R=$PWD; O=$R/gpurun_out/r6e; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | cut -c1-400) > $O/suite.log 2>&1
(timeout 600 python tools/probe/eimp_lockstep_diff.py 96 2>&1 | tail -40) > $O/lockstep_diff.log 2>&1
(timeout 900 python bench.py 2>$O/bench.err | tail -1) > $O/bench.json
tail -60 $O/suite.log | cut -c1-300; tail -5 $O/lockstep_diff.log
