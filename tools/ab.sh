#!/bin/bash
# tools/ab.sh libA.so libB.so [workload] — the same frame through two builds of
# the library, alternately, on one box (boxes differ by a few percent).
A=$1; B=$2; W=${3:-10k-spheres-4k-depth8}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for k in 1 2 3; do
  timeout -k 10 120 python3 $ROOT/tools/stage_times.py $A $W || exit 1
  timeout -k 10 120 python3 $ROOT/tools/stage_times.py $B $W || exit 1
done
