#!/usr/bin/env python3
"""Prints how much work the sphere hierarchy does per bounce ray on a workload
(self-checking kernels; slow).  Usage: python tools/walk_stats.py [workload]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
import torch, rtamd, bench
name = sys.argv[1] if len(sys.argv) > 1 else "10k-spheres-4k-depth8"
sc, w, h, depth = bench.build_scene(name)
fb = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
stats = torch.zeros(16, dtype=torch.int64, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
with rtamd.DeviceScene(sc) as ds:
    ds.render_device(fb.data_ptr(), None, w, h, max_depth=depth, dev_ray_count=cnt.data_ptr(),
                     dev_selfcheck_stats=stats.data_ptr())
    torch.cuda.synchronize()
s = [int(x) for x in stats.cpu()]
print("rays", int(cnt.item()), "screen: dropped-but-candidate", s[0], "kept", s[1])
for label, o in (("spheres nearest", 2), ("spheres shadow", 5), ("triangles nearest", 8),
                 ("triangles shadow", 11)):
    r = max(1, s[o])
    print(f"{label}: walks {s[o]}  child bounds/ray {s[o+1]/r:.1f}  leaf tests/ray {s[o+2]/r:.2f}")
