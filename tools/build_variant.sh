#!/bin/bash
# Build an A/B variant of libpdehip.so: only the stencil translation unit is recompiled with extra flags, the other objects are
# those of the regular build.  usage: tools/build_variant.sh <name> <extra hipcc flags...>   ->  tools/variants/libpdehip_<name>.so
# (git-ignored; travels to the GPU box; selected there with PDEHIP_LIB, see tools/ab_time.sh)
set -e
cd "$(dirname "$0")/../py-pde_amd"
name=$1; shift
mkdir -p ../tools/variants build_$name
make -s build/pdehip_sources.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Ibuild "$@" -c csrc/pdehip_kernels.hip -o build_$name/pdehip_kernels.o
objs=$(ls build/*.o | grep -v pdehip_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/variants/libpdehip_$name.so build_$name/pdehip_kernels.o $objs -ldl
ls -la ../tools/variants/libpdehip_$name.so
