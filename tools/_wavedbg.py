import os, sys
REPO = "/root/repo" if os.path.exists("/root/repo") else os.getcwd()
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
import torch, rtamd, bench, ctypes as C, numpy as np
sc, w, h, depth = bench.build_scene("10k-spheres-4k-depth8")
fb = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
L = rtamd.lib(); L.rt_debug_waves.argtypes = [C.c_void_p, C.c_int]
for n in (8, 1):
  for level in (2, 4):
    dbg = torch.zeros(4 * 8192, dtype=torch.int64, device="cuda")
    with rtamd.DeviceScene(sc) as ds:
        for rep in range(2):
            dbg.zero_()
            L.rt_debug_waves(dbg.data_ptr(), level)
            ds.render_device(fb.data_ptr(), None, w, h, max_depth=depth, band_rows=8, band_index=0, band_count=n)
            torch.cuda.synchronize()
    L.rt_debug_waves(None, 2)
    d = dbg.cpu().numpy().reshape(8192, 4)
    live = d[:, 1] > 0
    t0, t1 = d[live, 0], d[live, 1]
    it = (d[live, 2] & 0xffffffff); rf = d[live, 2] >> 32; rays = d[live, 3]
    base = t0.min(); span = (t1.max() - base) / 100.0
    dur = (t1 - t0) / 100.0
    print(f"share 1/{n} level {level}: waves {live.sum()} rays {rays.sum()} kernel span {span:.1f} us | wave start: median {np.median(t0-base)/100:.1f} max {(t0.max()-base)/100:.1f} us | wave duration: median {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} max {dur.max():.1f} us | iterations: median {np.median(it):.0f} p90 {np.percentile(it,90):.0f} max {it.max()} | us/iter median {np.median(dur/np.maximum(it,1)):.2f} | rays/wave median {np.median(rays):.0f} max {rays.max()}", flush=True)
