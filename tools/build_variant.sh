#!/bin/bash
# builds imp-release_amd/csrc/variants/libimp_hip_<tag>.so with extra compiler flags (kernel A/B switches are -D macros):
#   tools/build_variant.sh pkadd -DPP_PKADD=1     then     IMP_HIP_LIB=$PWD/imp-release_amd/csrc/variants/libimp_hip_pkadd.so python bench.py
set -e
TAG=$1; shift
C=imp-release_amd/csrc; V=$C/variants/$TAG; mkdir -p $V
for f in gemm_f32 gemm_wf superpoint attention_f32 attention_f16x3 ot ot_resident pool_misc pose context; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $C/$f.hip -o $V/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $V/*.o -o $C/variants/libimp_hip_$TAG.so
rm -rf $V
ls -la $C/variants/libimp_hip_$TAG.so
