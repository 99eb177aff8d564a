#!/bin/bash
# tools/profile.sh <tag> [bench.py args...] — rocprofv3 passes over bench.py on
# the GPU box.  Writes under gpurun_out/prof_<tag>/; copy the summaries you want
# to keep into profiles/.  Each pass is its own run (counters are never combined
# with tracing), each under its own time limit.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline $*"

timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- python3 $ARGS > "$OUT/stats.log" 2>&1 || { echo "stats pass failed"; tail -5 "$OUT/stats.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY -d "$OUT/pmc_sq" -o pmc -- python3 $ARGS > "$OUT/pmc_sq.log" 2>&1 || { echo "sq pass failed"; tail -5 "$OUT/pmc_sq.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d "$OUT/pmc_sq2" -o pmc -- python3 $ARGS > "$OUT/pmc_sq2.log" 2>&1 || { echo "sq2 pass failed"; tail -5 "$OUT/pmc_sq2.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_write" -o pmc -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }

python3 $ROOT/tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
