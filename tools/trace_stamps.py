"""Where the cycles of a k_trace wave go (diagnostic build: make stamps).
Renders the bench frame with all bounce levels in one launch and prints the stamp
counters of k_trace (rt_trace.h)."""
import ctypes as C
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import rtamd  # noqa: E402
rtamd.LIB_PATH = os.path.join(REPO, "build", "librt_amd_stamps.so")
import bench  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "10k-spheres-4k-depth8"
sc, w, h, depth = bench.build_scene(name)
dev = torch.device("cuda", 0)
fb = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
L = rtamd.lib()
out = (C.c_ulonglong * 24)()
with rtamd.DeviceScene(sc) as ds:
    ds.set_one_launch(int(os.environ.get("ONE_LAUNCH", "1")))
    for k in range(4):
        ds.render_device(fb.data_ptr(), None, w, h, max_depth=depth, band_rows=8, band_index=0,
                         band_count=int(os.environ.get("SHARE", "1")))
        torch.cuda.synchronize()
        L.rt_debug_trace_stamps(out)
v = [float(x) for x in out]
wv = max(1.0, v[12])
tot = max(1.0, v[0])
print(f"waves {v[12]:.0f}, cycles per wave {v[0] / wv:.0f}")
for k, nm in ((1, "node visits"), (2, "parked leaves"), (3, "handing in"), (4, "shading"), (5, "fetching")):
    print(f"  {nm:14s} {100 * v[k] / tot:5.1f} %")
print(f"per wave: {v[6] / wv:.0f} visits ({v[1] / max(1, v[6]):.0f} cycles each, {v[11] / max(1, v[6]):.1f} lanes walking), "
      f"{v[7] / wv:.0f} service phases ({v[13] / wv:.0f} forced by full parking rows)")
print(f"per service phase: {v[8] / max(1, v[7]):.2f} screen batches ({v[15] / max(1, v[8]):.1f} lanes), "
      f"{v[9] / max(1, v[7]):.2f} exact rounds ({v[14] / max(1, v[9]):.1f} lanes), "
      f"{v[10] / max(1, v[7]):.2f} batches shaded")
print(f"cycles per service phase: leaves {v[2] / max(1, v[7]):.0f}, hand-in {v[3] / max(1, v[7]):.0f}, "
      f"shading {v[4] / max(1, v[7]):.0f} ({v[4] / max(1, v[10]):.0f} per batch), fetch {v[5] / max(1, v[7]):.0f}")
first, last = v[19], v[17]
mean_end = (v[18] / wv) - (int(first) & 0xffffffffff)
print(f"longest wave {v[16]:.0f} cycles; last wave ends {last - first:.0f} cycles after the first one starts, "
      f"the mean wave at {mean_end:.0f} ({100 * mean_end / max(1.0, last - first):.0f} % of that)")
print(f"waves that never visited a node: {v[20]:.0f}; waves that lived under 10 M cycles: {v[21]:.0f}")
print(f"lane-visits of nearest-hit rays: {v[22]:.0f}, of which into a node already beyond the best hit: {v[23]:.0f} ({100 * v[23] / max(1.0, v[22]):.1f} %)")
