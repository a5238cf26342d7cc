#!/bin/bash
# builds a copy of the library with extra -D flags for A/B runs: tools/build_variant.sh NAME -DX=1 ...
# -> seqkit_amd/variants/libseqkit_hip_NAME.so (load it with SKH_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p seqkit_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -I include -I seqkit_amd/csrc -Wall -Wno-unused-function \
  -o seqkit_amd/variants/libseqkit_hip_$name.so seqkit_amd/csrc/capi.hip seqkit_amd/csrc/kernels.hip seqkit_amd/csrc/parse.hip \
  seqkit_amd/csrc/format.hip seqkit_amd/csrc/slices.hip seqkit_amd/csrc/gaps.hip seqkit_amd/csrc/multi.hip seqkit_amd/csrc/shares.hip -ldl -lpthread
echo built seqkit_amd/variants/libseqkit_hip_$name.so
