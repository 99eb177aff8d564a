#!/bin/bash
# tools/pmc_sq.sh [bench args] — instruction and wave counters of the library's
# kernels (one frame in flight, no side stream); per-kernel means.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-plain-loop --frames-in-flight 1 --side-stream 0 $*"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/sq -o pmc -- python3 $ARGS > $OUT/sq.log 2>&1 || { echo "sq failed"; tail -3 $OUT/sq.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INST_CYCLES_VMEM -d $OUT/sq2 -o pmc -- python3 $ARGS > $OUT/sq2.log 2>&1 || { echo "sq2 failed"; tail -3 $OUT/sq2.log; exit 1; }
python3 - $OUT <<'PY'
import sqlite3, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for db in glob.glob(sys.argv[1] + "/*/*.db"):
    con = sqlite3.connect(db)
    for k, c, v in con.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like '%rt::k_%'"):
        agg[k[:50]][c].append(v)
for k in sorted(agg):
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    n = len(next(iter(agg[k].values())))
    print("%-50s n=%2d valu %.3g salu %.3g vmem_rd %.3g vmem_wr %.3g | lanes/valu %.1f | wave_cyc %.3g wait_any %.0f%% wait_inst %.0f%% | gui/8 %.3g cyc" % (
        k, n, c.get('SQ_INSTS_VALU', 0), c.get('SQ_INSTS_SALU', 0), c.get('SQ_INSTS_VMEM_RD', 0), c.get('SQ_INSTS_VMEM_WR', 0),
        c.get('SQ_THREAD_CYCLES_VALU', 0) / max(1, c.get('SQ_ACTIVE_INST_VALU', 1)) / 4 * 4,
        c.get('SQ_WAVE_CYCLES', 0), 100 * c.get('SQ_WAIT_ANY', 0) / max(1, c.get('SQ_WAVE_CYCLES', 1)),
        100 * c.get('SQ_WAIT_INST_ANY', 0) / max(1, c.get('SQ_WAVE_CYCLES', 1)), c.get('GRBM_GUI_ACTIVE', 0) / 8))
PY
