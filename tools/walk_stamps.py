"""Where the cycles of a walking wave go (diagnostic build: make stamps).
Renders the bench frame with a launch per stage and level and prints the stamp
counters of k_walk<false,false> (rt_walk.h)."""
import ctypes as C
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import rtamd  # noqa: E402
rtamd.LIB_PATH = os.path.join(REPO, "build", os.environ.get("STAMPS_LIB", "librt_amd_stamps.so"))
import bench  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "10k-spheres-4k-depth8"
sc, w, h, depth = bench.build_scene(name)
dev = torch.device("cuda", 0)
fb = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
L = rtamd.lib()
out = (C.c_ulonglong * 16)()
with rtamd.DeviceScene(sc) as ds:
    ds.set_one_launch(False)
    ds.set_side_stream(False)
    for k in range(3):
        ds.render_device(fb.data_ptr(), None, w, h, max_depth=depth)
        torch.cuda.synchronize()
        L.rt_debug_stamps(out)
v = list(out)
waves = max(1, v[6])
print(f"waves {v[6]}  (per frame: {v[6]} wave launches of k_walk<false,false>)")
print(f"cycles per wave: total {v[0]/waves:.0f}  service {v[1]/waves:.0f}  node-load wait {v[2]/waves:.0f}  rest of visits {v[3]/waves:.0f}")
print(f"service phases {v[4]/waves:.1f} per wave, {v[1]/max(1,v[4]):.0f} cycles each, of which cursor+item wait {v[7]/max(1,v[4]):.0f}")
print(f"visits {v[5]/waves:.1f} per wave, load wait {v[2]/max(1,v[5]):.0f} + other {v[3]/max(1,v[5]):.0f} cycles each")
print(f"visits with an on-the-spot leaf test: {100.0 * v[8] / max(1, v[5]):.1f} %; lanes with a ray per visit {v[9] / max(1, v[5]):.1f}")
print(f"of the service time, testing parked leaves: {100.0 * v[10] / max(1, v[1]):.0f} %")
