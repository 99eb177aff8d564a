"""Rays traced per max_depth on a bench workload (differences = rays of each
bounce level)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
sys.path.insert(0, REPO)
import torch
import rtamd, bench
name = sys.argv[1] if len(sys.argv) > 1 else "10k-spheres-4k-depth8"
sc, w, h, depth = bench.build_scene(name)
dev = torch.device("cuda", 0)
fb = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
prev = 0
with rtamd.DeviceScene(sc) as ds:
    for d in range(1, depth + 1):
        cnt.zero_()
        ds.render_device(fb.data_ptr(), None, w, h, max_depth=d, dev_ray_count=cnt.data_ptr())
        torch.cuda.synchronize()
        n = int(cnt.item())
        print(f"depth {d}: {n} rays (+{n - prev})")
        prev = n
