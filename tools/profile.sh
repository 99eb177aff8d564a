#!/bin/bash
# tools/profile.sh <tag> <workload> [bench.py args...] — rocprofv3 passes over
# bench.py on the GPU box, ONE frame in flight so that no dispatch is stretched by
# another frame's kernels.  Writes under gpurun_out/prof_<tag>/ and updates
# gpurun_out/prof_<tag>/counters.json; copy summary.txt and that entry into
# profiles/.  Each pass is its own run (counters are never combined with
# tracing), each under its own time limit.
set -u
TAG=$1; WORKLOAD=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ONE_LAUNCH=1
for a in "$@"; do [ "$a" = "--one-launch=0" ] && ONE_LAUNCH=0; done
ARGS="$ROOT/bench.py --workload $WORKLOAD --steps 5 --warmup 2 --no-cpu-baseline --no-plain-loop --frames-in-flight 1 --side-stream 0 $*"

pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 "$@" -d "$OUT/$name" -o $name -- python3 $ARGS > "$OUT/$name.log" 2>&1 || { echo "$name pass failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass stats --kernel-trace --stats
pass pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass pmc_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum

python3 $ROOT/tools/summarize_prof.py "$OUT" "$WORKLOAD" "$OUT/counters.json" $ONE_LAUNCH > "$OUT/summary.txt" 2>&1
grep -A40 "== per frame" "$OUT/summary.txt"
grep '"metric"' "$OUT/stats.log" > "$OUT/bench_line_under_rocprof.json"
# the passes' databases are tens of megabytes each and summarised above: only the
# text travels back
for d in stats pmc_sq pmc_sq2 pmc_lds pmc_fetch pmc_write; do rm -rf "$OUT/$d"; done
