#!/bin/bash
# tools/rank_share_trace.sh N — kernel trace of one rank's share of an N-GPU frame
# (tools/rank_share.py): per-dispatch start and duration of the last frame.
set -u
N=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/share_$N
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RANK_SHARE_N=$N
timeout -s KILL 300 rocprofv3 --kernel-trace -d "$OUT" -o kt -- python3 $ROOT/tools/rank_share.py > "$OUT/run.log" 2>&1 || { echo failed; tail -5 "$OUT/run.log"; exit 1; }
python3 - "$OUT" <<'PY'
import sqlite3, glob, sys
con = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
rows = [r for r in con.execute("select name, start, end, stream_id from kernels order by start") if "rt::" in r[0]]
idx = [i for i, r in enumerate(rows) if "k_render" in r[0]]
first = idx[-1]
t0 = rows[first][1]
busy = 0
for n, s, e, st in rows[first:]:
    print("%8.3f ms  +%7.3f ms  stream %s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, st, n[:44]))
print("frame span %.3f ms" % ((rows[-1][2] - t0) / 1e6))
PY
