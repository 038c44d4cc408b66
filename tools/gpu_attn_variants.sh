#!/bin/bash
# runs every tools/probe/attn_<tag>.bin (timing, 4 x 100 launches at B = 4, N = 2048) and attn_<tag>_prof.bin (phase stamps + per-SIMD timeline);
# usage: gpurun -- bash tools/gpu_attn_variants.sh OUTNAME tag1 tag2 ...
OUT=gpurun_out/${1:-attn_variants}.log; shift
mkdir -p gpurun_out; : > $OUT
for rep in 1 2; do
  for tag in "$@"; do
    echo "=== $tag (timing, pass $rep)" >> $OUT
    timeout 60 tools/probe/attn_$tag.bin 4 2048 1 2>&1 | tail -3 >> $OUT
  done
done
for tag in "$@"; do
  echo "=== $tag (phase stamps)" >> $OUT
  timeout 60 tools/probe/attn_${tag}_prof.bin 4 2048 1 2>&1 | tail -32 >> $OUT
done
cat $OUT
