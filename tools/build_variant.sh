#!/bin/bash
# usage: tools/build_variant.sh <name> [-DMACRO ...]   -- an experimental build of the library into _ab/<name>.so (MNW_LIB_PATH picks it up;
# tools/ab_libs.sh compares several in one GPU call).  Not the product build: that is minnow_amd/build.py.
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p _ab
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -fno-fast-math -x hip -Wno-unused-function -Wno-unused-value -Wno-unused-result "$@" \
  minnow_amd/csrc/mnw_kernels.hip minnow_amd/csrc/mnw_api.cpp -o _ab/$name.so && echo "_ab/$name.so"
