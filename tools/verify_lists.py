#!/usr/bin/env python3
"""rt_verify_sphere_lists on a bench workload: every sphere against every pixel ray and
primary shadow ray of the frame (slow), counting spheres that pass the reference's
discriminant test but are missing from the ray's list (must be 0).
usage: verify_lists.py [workload]"""
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
import rtamd  # noqa: E402
import bench  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "10k-spheres-4k-depth1"
sc, w, h, depth = bench.build_scene(name)
with rtamd.DeviceScene(sc) as ds:
    t = time.time()
    served, missing, s_served, s_missing = ds.verify_sphere_lists(w, h)
    d = ds.describe()
print(f"{name}: {served} of {w * h} pixel rays served by a list, {missing} passing spheres missing; "
      f"{s_served} shadow rays served, {s_missing} missing; {time.time() - t:.1f} s; "
      f"{d['screen_cells_with_list']} screen cells with a list, "
      f"{d['light_cells_without_list']} light cells without")
