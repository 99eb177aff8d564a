#!/bin/bash
# tools/kernel_overlap.sh [bench args] — kernel trace of one bench run; prints the
# last frame's dispatches with start / end relative to the frame's first kernel
# and the stream they ran on, to see which launches overlap.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/overlap
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d "$OUT" -o kt -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --screen bvh "$@" > "$OUT/run.log" 2>&1 || { echo "failed"; tail -5 "$OUT/run.log"; exit 1; }
python3 - "$OUT" <<'PY'
import sqlite3, glob, sys
db = glob.glob(sys.argv[1] + "/*.db")[0]
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
# last frame = from the last k_render-ish group start
rows = [r for r in rows if "rt::" in r[0] and "k_bounce" not in r[0] and "k_render<0" not in r[0]]
idx = [i for i, r in enumerate(rows) if "k_render" in r[0]]
first = idx[-1]
while first > 0 and "k_render" in rows[first - 1][0]:
    first -= 1
t0 = rows[first][1]
for n, s, e, st, q in rows[first:]:
    if True:
        print("%9.3f -> %9.3f ms  stream %s queue %s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, st, q, n[:40]))
PY
