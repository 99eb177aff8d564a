#!/usr/bin/env python3
"""Condense the rocprofv3 result databases written by tools/profile.sh into one
text summary: per-kernel time statistics and per-kernel mean counter values.
Usage: summarize_prof.py <dir>"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
dbs = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True))
for db in dbs:
    con = sqlite3.connect(db)
    rel = os.path.relpath(db, root)
    try:
        rows = list(con.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels"))
    except sqlite3.Error:
        rows = []
    n_pmc = con.execute("select count(*) from counters_collection").fetchone()[0]
    if rows and not n_pmc:
        print(f"== kernel time (rocprofv3 --kernel-trace --stats): {rel}")
        print(f"   {'kernel':72s} {'calls':>6s} {'total_ms':>11s} {'avg_ms':>10s} {'%':>6s}")
        for name, calls, total, avg, pct in rows:
            # the top_kernels view reports microseconds
            print(f"   {name[:72]:72s} {calls:6d} {total/1e3:11.3f} {avg/1e3:10.4f} {pct:6.2f}")
    if n_pmc:
        print(f"== counters (rocprofv3 --pmc), mean per dispatch: {rel}")
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
             "group by kernel_name, counter_name order by kernel_name, counter_name")
        last = None
        for k, c, n, v in con.execute(q):
            if "rt::" not in k:
                continue
            if k != last:
                print("   kernel:", k[:110])
                last = k
            print(f"      {c:24s} dispatches={n:3d} mean={v:.6g}")
