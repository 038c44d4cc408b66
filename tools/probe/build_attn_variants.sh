#!/bin/bash
# builds tools/probe/attn_<tag>.bin (timing) and attn_<tag>_prof.bin (phase stamps + timeline) for a list of "tag:flags" variants of
# csrc/attention_f16x3.hip;   usage: tools/probe/build_attn_variants.sh "base:" "spread:-DPP_SPREAD=1" ...
cd "$(dirname "$0")"
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function $flags attn_probe.hip -o attn_$tag.bin &
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -DPP_PROFILE -DPP_TIMELINE $flags attn_probe.hip -o attn_${tag}_prof.bin &
done
wait
ls -la attn_*.bin
