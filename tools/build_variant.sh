#!/bin/bash
# Builds a variant of the library for A/B measurements: tools/build_variant.sh <name> [extra compiler flags...]
# -> expl/librt_<name>.so (expl/ is not tracked; it travels to the GPU box with the tree)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/expl"
C=$ROOT/raytracing_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize \
  -Wno-unused-function -Wno-unused-value -Wno-unused-result "$@" \
  -o "$ROOT/expl/librt_$NAME.so" $C/rt_device.hip $C/rt_gpubuild.hip $C/rt_bvh.cpp $C/rt_host.cpp $C/rt_scene.cpp $C/rt_accel.cpp -lpthread
echo "built expl/librt_$NAME.so"
