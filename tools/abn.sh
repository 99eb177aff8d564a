#!/bin/bash
# tools/abn.sh <workload> <rounds> name1 name2 ... — the same frame through several
# builds of the library (build/<name>/librt_amd.so, tools/build_variant.sh),
# alternately, on one box (boxes differ by a few percent).
W=$1; R=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for k in $(seq 1 $R); do
  for n in "$@"; do
    timeout -k 10 120 python3 $ROOT/tools/stage_times.py $ROOT/build/$n/librt_amd.so $W || exit 1
  done
done
