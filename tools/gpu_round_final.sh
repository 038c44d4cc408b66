#!/bin/bash
# round 6: full GPU suite on the committed tree, the round's profile evidence, the driver's command
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider 2>&1 | cut -c1-400) > $O/suite.log 2>&1
bash tools/gpu_profile_round.sh r6 > $O/profile_round.log 2>&1
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | tail -1) > $O/bench_driver_command.json
tail -25 $O/suite.log | cut -c1-300
