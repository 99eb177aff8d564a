#!/bin/bash
# tools/kernel_times.sh <tag> [bench args] — rocprofv3 kernel trace of one bench
# run; prints per-dispatch durations of the library's kernels in launch order.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kt_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT" -o kt -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 --side-stream 0 "$@" > "$OUT/run.log" 2>&1 || { echo "failed"; tail -5 "$OUT/run.log"; exit 1; }
python3 - "$OUT" <<'PY'
import sqlite3, glob, sys
db = glob.glob(sys.argv[1] + "/*.db")[0]
con = sqlite3.connect(db)
cur = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
for r in cur:
    print("%-90s calls=%4d total_us=%10.1f avg_us=%10.1f %5.1f%%" % (r[0][:90], r[1], r[2]/1e3 if r[2] > 1e6 else r[2], r[3]/1e3 if r[3] > 1e6 else r[3], r[4]))
cols = [d[1] for d in con.execute("pragma table_info(kernels)")]
print(cols)
q = "select name, start, end from kernels order by start"
rows = list(con.execute(q))
t0 = rows[0][1]
for n, s, e in rows[-40:]:
    if "rt::" in n:
        print("%10.3f ms  +%9.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n[:60]))
PY
