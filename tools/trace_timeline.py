"""Service-phase timeline of a few waves of k_trace (diagnostic build: make stamps).
usage: SHARE=8 trace_timeline.py [workload]  -> per sampled wave, one line per service
phase: time, lanes walking, items on the wave's stack, hits waiting, visits since the
last phase."""
import ctypes as C
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rtamd  # noqa: E402
rtamd.LIB_PATH = os.path.join(REPO, "build", "librt_amd_stamps.so")
import bench  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "10k-spheres-4k-depth8"
sc, w, h, depth = bench.build_scene(name)
fb = torch.zeros(h, w, 4, dtype=torch.uint8, device="cuda")
L = rtamd.lib()
nw, nr = C.c_int(), C.c_int()
buf = np.zeros(32 * 1024 * 4, dtype=np.uint32)
with rtamd.DeviceScene(sc) as ds:
    for k in range(3):
        ds.render_device(fb.data_ptr(), None, w, h, max_depth=depth, band_rows=8, band_index=0,
                         band_count=int(os.environ.get("SHARE", "1")))
        torch.cuda.synchronize()
        L.rt_debug_trace_log(buf.ctypes.data_as(C.c_void_p), C.byref(nw), C.byref(nr))
log = buf.reshape(nw.value, nr.value, 4)
show = int(os.environ.get("SHOW", "3"))
for wv in range(nw.value):
    recs = log[wv]
    n = int((recs[:, 0] > 0).sum()) + 1
    recs = recs[:n]
    if n <= 1:
        continue
    t = recs[:, 0].astype(np.float64) * 16 / 2.4e3  # us at 2.4 GHz
    busy, spw = recs[:, 1], recs[:, 2]
    hits, drained, visits = recs[:, 3] & 0xffff, (recs[:, 3] >> 16) & 1, recs[:, 3] >> 20
    first_drained = int(np.argmax(drained)) if drained.any() else -1
    print(f"wave sample {wv}: {n} service phases, last at {t[-1]:.0f} us, queue drained from phase "
          f"{first_drained} ({t[first_drained]:.0f} us), visits {int(visits.sum())}, "
          f"mean lanes walking at phase entry {busy.mean():.1f}")
    if wv < show:
        for k in range(n):
            print(f"   {t[k]:8.1f} us  walking {busy[k]:2d}  stack {spw[k]:5d}  hits {hits[k]:3d}  "
                  f"drained {drained[k]}  visits since {visits[k]}")
