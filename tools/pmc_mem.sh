#!/bin/bash
# tools/pmc_mem.sh [bench args] — memory-path counters of the library's kernels
# (one rocprofv3 --pmc pass per counter group, one frame in flight, no side
# stream so that every dispatch runs alone).  Prints per-kernel means.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_mem
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-plain-loop --frames-in-flight 1 --side-stream 0 $*"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/$name -o pmc -- python3 $ARGS > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -3 $OUT/$name.log; exit 1; }
}
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
pass ta TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum GRBM_GUI_ACTIVE
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_ANY
python3 - $OUT <<'PY'
import sqlite3, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for db in glob.glob(root + "/*/*.db"):
    con = sqlite3.connect(db)
    for k, c, v in con.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like '%rt::k_%'"):
        agg[k[:60]][c].append(v)
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-34s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
