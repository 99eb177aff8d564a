#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags] — the library of the current
# tree as build/<name>/librt_amd.so, for tools/ab.sh (two builds alternately on
# one box).  Never loaded by the product path.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build/$NAME"
C=$ROOT/ray-tracer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-fast-math \
  -Wall -Wno-unused-function "$@" -shared -o "$ROOT/build/$NAME/librt_amd.so" \
  $C/rt_kernels.hip $C/rt_api.hip $C/rt_bvh.cpp $C/rt_tribvh.cpp
echo "built build/$NAME/librt_amd.so"
