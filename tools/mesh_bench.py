#!/usr/bin/env python3
"""Times the triangle-mesh configuration: scene build (upload + hierarchy) and
frames through rt_render_device.  usage: mesh_bench.py NX NY GRID W H DEPTH [ROUGH]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import rtamd  # noqa: E402
import torch  # noqa: E402

nx, ny, grid, w, h, depth = (int(x) for x in sys.argv[1:7])
rough = float(sys.argv[7]) if len(sys.argv) > 7 else 0.0
t = time.perf_counter()
sc = rtamd.mesh_scene(nx, ny, grid, roughness=rough)
print(f"scene: {sc.t_size} triangles, {sc.s_size} spheres ({time.perf_counter() - t:.2f} s to generate)", flush=True)
t = time.perf_counter()
ds = rtamd.DeviceScene(sc)
print(f"rt_scene_create: {time.perf_counter() - t:.2f} s", flush=True)
dev = torch.device("cuda", 0)
fb = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for k in range(3):
    cnt.zero_()
    torch.cuda.synchronize()
    t = time.perf_counter()
    ds.render_device(fb.data_ptr(), None, w, h, max_depth=depth, stream=stream, dev_ray_count=cnt.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    n = int(cnt.item())
    print(f"frame {k}: {dt * 1e3:.1f} ms, {n} rays, {n / dt / 1e6:.1f} Mrays/s, overflow={ds.queue_overflowed()}", flush=True)
print("fb checksum", int(fb.to(torch.int64).sum().item()))
ds.close()
