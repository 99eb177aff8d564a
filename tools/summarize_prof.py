#!/usr/bin/env python3
"""Condense the rocprofv3 result databases written by tools/profile.sh.

Prints, per pass, the per-kernel time statistics or mean counter values, then a
per-frame table by stage (primary = k_render, bounce = k_trace or the k_walk /
k_shade chain, compose = k_compose), and with a third argument updates that
workload's entry in a counters file (profiles/counters.json) that bench.py
reads for its roofline figures.

FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per
128-byte request of a wide read, so it is doubled.  SQ_WAVE_CYCLES and the
SQ_WAIT_* / SQ_ACTIVE_INST_* counters tick once per four cycles.

Usage: summarize_prof.py <dir> [workload counters.json one_launch]"""
import collections
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

root = sys.argv[1]
STAGE_OF = (("k_render<", "primary"), ("k_trace", "bounce"), ("k_walk<", "bounce"),
            ("k_shade", "bounce"), ("k_bounce<", "bounce"), ("k_compose", "compose"))


def stage_of(kernel):
    for key, st in STAGE_OF:
        if "rt::" + key in kernel:
            return st
    return None


def short(kernel):
    k = kernel.replace("void ", "")
    return k[:k.index("(")] if "(" in k else k


dbs = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True))
per_frame = collections.defaultdict(lambda: collections.defaultdict(float))  # stage -> name -> sum
kernels_of = collections.defaultdict(set)
frames_seen = {}

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
        # per frame and stage: durations of the dispatches, in launch order
        n_frames = 0
        for name, s, e in con.execute("select name, start, end from kernels order by start"):
            st = stage_of(name)
            if st is None:
                continue
            if st == "primary":
                n_frames += 1
            per_frame[st]["ms"] += (e - s) / 1e6
            kernels_of[st].add(short(name))
        frames_seen["ms"] = n_frames
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
        n_frames = collections.Counter()
        q = "select kernel_name, counter_name, value from counters_collection"
        for k, c, v in con.execute(q):
            st = stage_of(k)
            if st is None:
                continue
            per_frame[st][c] += v
            if st == "primary":
                n_frames[c] += 1
        for c, n in n_frames.items():
            frames_seen[c] = n

print("== per frame, by stage (sums over the stage's dispatches / frames profiled)")
stages = {}
for st in ("primary", "bounce", "compose"):
    if st not in per_frame:
        continue
    v = {c: x / max(1, frames_seen.get(c, 1)) for c, x in per_frame[st].items()}
    ms = v.get("ms", 0.0)
    insts = v.get("SQ_INSTS_VALU", 0.0)
    lanes = v.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(1.0, v.get("SQ_ACTIVE_INST_VALU", 1.0))
    wave_cyc = max(1.0, v.get("SQ_WAVE_CYCLES", 1.0))
    gui = v.get("GRBM_GUI_ACTIVE", 0.0) / 8
    # the clock the chip held (counter of one pass over the duration of another:
    # reads high on short dispatches, so it is capped at the maximum clock)
    clock = min(2.4, gui / (ms * 1e-3) / 1e9) if ms > 0 and gui > 0 else 2.4
    hbm = (v.get("FETCH_SIZE", 0.0) * 2 + v.get("WRITE_SIZE", 0.0)) * 1024
    # mean resident waves per SIMD over the stage: wave-cycles over the cycles of the
    # launch on the chip's 1024 SIMDs.  As rocprofv3 sums it over the chip the counter
    # advances once per TWO cycles of a resident wave: calibrated on k_trace, whose
    # 5 120 waves (5 per SIMD, the launch bound) are resident for nearly the whole
    # launch and read 9.1 with a factor of 4; k_render (bound 4 per SIMD) and k_compose
    # (8) read 4.07 and 6.3 with 2
    occ = v.get("SQ_WAVE_CYCLES", 0.0) * 2 / max(1.0, ms * 1e-3 * clock * 1e9 * 1024)
    lds_active = v.get("SQ_LDS_IDX_ACTIVE", 0.0)
    lds_conflict = v.get("SQ_LDS_BANK_CONFLICT", 0.0)
    e = {
        "kernels": sorted(kernels_of[st]), "ms": ms, "valu_wave_insts": insts,
        "lanes_per_valu_inst": lanes, "clock_ghz": clock, "hbm_bytes": hbm,
        "fetch_size_kib": v.get("FETCH_SIZE", 0.0), "write_size_kib": v.get("WRITE_SIZE", 0.0),
        "waves": v.get("SQ_WAVES", 0.0), "wave_quad_cycles": v.get("SQ_WAVE_CYCLES", 0.0),
        "wait_any_frac": v.get("SQ_WAIT_ANY", 0.0) / wave_cyc,
        "wait_inst_any_frac": v.get("SQ_WAIT_INST_ANY", 0.0) / wave_cyc,
        "active_inst_any_frac": v.get("SQ_ACTIVE_INST_ANY", 0.0) / wave_cyc,
        "salu_insts": v.get("SQ_INSTS_SALU", 0.0), "smem_insts": v.get("SQ_INSTS_SMEM", 0.0),
        "vmem_rd_insts": v.get("SQ_INSTS_VMEM_RD", 0.0), "vmem_wr_insts": v.get("SQ_INSTS_VMEM_WR", 0.0),
        "lds_insts": v.get("SQ_INSTS_LDS", 0.0), "gui_active_cycles_per_xcd": gui,
        "occupancy_waves_per_simd": occ,
        "waves_per_busy_cycle": v.get("SQ_WAVE_CYCLES", 0.0) / max(1.0, v.get("SQ_BUSY_CYCLES", 1.0)),
        "lds_idx_active_cycles": lds_active, "lds_bank_conflict_cycles": lds_conflict,
        "lds_bank_conflict_frac": lds_conflict / lds_active if lds_active > 0 else None,
        "lds_addr_conflict_cycles": v.get("SQ_LDS_ADDR_CONFLICT", 0.0),
        "lds_wait_inst_frac": v.get("SQ_WAIT_INST_LDS", 0.0) / wave_cyc,
    }
    stages[st] = e
    lane_ops = insts * lanes
    print(f"   {st:8s} {'+'.join(e['kernels'])}")
    print(f"            {ms:8.3f} ms   VALU {insts:.4g} wave instructions x {lanes:.1f} lanes = "
          f"{lane_ops / max(ms, 1e-9) / 1e9:.2f} T lane-ops/s = "
          f"{lane_ops / max(ms, 1e-9) / 1e9 / 78.65:.3f} of 78.65;  issue "
          f"{insts * 2 / (1024 * clock * 1e6 * max(ms, 1e-9)):.3f} of 1 per 2 cycles and SIMD at {clock:.2f} GHz")
    print(f"            waves {e['waves']:.0f}, parked {e['wait_any_frac']:.2f}, issue-stalled "
          f"{e['wait_inst_any_frac']:.2f}, issuing {e['active_inst_any_frac']:.2f} of wave time;  "
          f"VMEM rd {e['vmem_rd_insts']:.3g} wr {e['vmem_wr_insts']:.3g}  LDS {e['lds_insts']:.3g}  "
          f"SALU {e['salu_insts']:.3g};  HBM {hbm / 1e6:.1f} MB ({hbm / max(ms, 1e-9) / 1e6:.0f} GB/s)")
    cf = e["lds_bank_conflict_frac"]
    print(f"            occupancy {occ:.2f} waves per SIMD (mean over the stage);  LDS: "
          f"{lds_active:.4g} active cycles, {lds_conflict:.4g} bank-conflict cycles"
          + (f" = {cf:.4f} of them" if cf is not None else " (pass not run)"))
total_ms = sum(e["ms"] for e in stages.values())
print(f"   frame: {total_ms:.3f} ms of kernel time ({frames_seen.get('ms', 0)} frames traced)")

if len(sys.argv) > 4:
    workload, path, one_launch = sys.argv[2], sys.argv[3], int(sys.argv[4])
    data = json.load(open(path)) if os.path.exists(path) else {}
    import bench
    data[workload] = {
        "source_sha": bench.kernel_source_sha(),   # the build these counts belong to
        "one_launch": one_launch, "frames_profiled": frames_seen.get("ms", 0),
        "frame_kernel_ms": total_ms, "stages": stages,
        "how": "tools/profile.sh: rocprofv3 --kernel-trace --stats, then separate --pmc passes "
               "(SQ, SQ2+GRBM, LDS, FETCH_SIZE, WRITE_SIZE) of bench.py --frames-in-flight 1; sums "
               "over a stage's dispatches divided by the frames profiled"}
    json.dump(data, open(path, "w"), indent=1)
    print(f"   -> {path} [{workload}]")
