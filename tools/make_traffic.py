#!/usr/bin/env python3
"""Reads the FETCH_SIZE / WRITE_SIZE passes written by tools/profile.sh and
prints HBM bytes per frame for the work-avoiding kernels of one workload.
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per
128-B request of a wide coalesced read, so it is doubled.
Usage: make_traffic.py <prof dir> <workload name> [traffic.json to update]"""
import glob, json, os, sqlite3, sys

root, workload = sys.argv[1], sys.argv[2]
FAST = ("k_render<3", "k_render<1", "k_walk<", "k_shade", "k_compose")


def per_frame(db, counter):
    con = sqlite3.connect(db)
    total, frames = 0.0, 0
    q = "select kernel_name, value from counters_collection where counter_name = ?"
    for k, v in con.execute(q, (counter,)):
        if any(f in k for f in FAST):
            total += v
            if "k_render<" in k:
                frames += 1
    return total / max(1, frames), frames

fetch, nf = per_frame(glob.glob(os.path.join(root, "pmc_fetch", "*.db"))[0], "FETCH_SIZE")
write, nw = per_frame(glob.glob(os.path.join(root, "pmc_write", "*.db"))[0], "WRITE_SIZE")
out = {"fetch_size_kib_per_frame": fetch, "write_size_kib_per_frame": write,
       "hbm_bytes_per_frame": int(fetch * 1024 * 2 + write * 1024), "frames_profiled": nf,
       "note": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and WRITE_SIZE in separate passes, summed "
               "over the kernels of one rt_render_device call"}
print(json.dumps({workload: out}, indent=1))
if len(sys.argv) > 3:
    path = sys.argv[3]
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[workload] = out
    json.dump(data, open(path, "w"), indent=1)
