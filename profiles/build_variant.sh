#!/bin/bash
# Builds an experimental copy of the product library with extra compiler flags on the kernel file:
#   bash profiles/build_variant.sh <name> [-DRT_EXP_...=1 ...]   ->  build_abl/<name>.so
# (build_abl/ is git-ignored; RTIOW_LIB=./build_abl/<name>.so selects it for a run)
set -e
cd "$(dirname "$0")/../raytracing-in-one-weekend_amd/csrc"
name=$1; shift
mkdir -p ../../build_abl
make -s host_scene.o fast_bvh.o image_decode.o device_bvh.o rt_capi.o rt_group.o
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -O1 --offload-arch=gfx950 "$@" -c rt_kernels.hip -o ../../build_abl/$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_abl/$name.so host_scene.o fast_bvh.o image_decode.o ../../build_abl/$name.o device_bvh.o rt_capi.o rt_group.o -ldl
rm -f ../../build_abl/$name.o
echo build_abl/$name.so
