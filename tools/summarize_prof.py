#!/usr/bin/env python3
"""Condense rocprofv3 output directories written by tools/profile.sh into one
text summary: per-kernel time statistics and per-kernel mean counter values."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


for f in find("*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, root))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            print("  " + ", ".join(row))
            if i > 12:
                break

for f in find("*counter_collection.csv"):
    print("== counters:", os.path.relpath(f, root))
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        if "k_render" not in k and "k_bounce" not in k:
            continue
        print("  kernel:", k[:100])
        for c, vals in sorted(cs.items()):
            print(f"    {c:28s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g}")
