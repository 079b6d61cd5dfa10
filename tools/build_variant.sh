#!/bin/bash
# Builds a variant of the library with extra device-code defines (tuning only):
#   tools/build_variant.sh name -DKC_UP_RU=2 -DKC_UP_HARDWIRE   ->  profiles/ab_libs/name.so   (git-ignored, travels with gpurun)
set -eu
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/kanter_core_amd/csrc/build
mkdir -p $R/profiles/ab_libs /tmp/kc_variant_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-fast-math -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"
/opt/rocm/bin/hipcc $F -mllvm -structurizecfg-skip-uniform-regions=1 -x hip "$@" -c $R/kanter_core_amd/csrc/kernels.hip -o /tmp/kc_variant_$name/kernels.o
/opt/rocm/bin/hipcc $F "$@" -c $R/kanter_core_amd/csrc/resize.cpp -o /tmp/kc_variant_$name/resize.o
objs=$(ls $B/*.o | grep -v -e kernels.o -e resize.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/profiles/ab_libs/$name.so /tmp/kc_variant_$name/kernels.o /tmp/kc_variant_$name/resize.o $objs -lz -ldl
echo $R/profiles/ab_libs/$name.so
