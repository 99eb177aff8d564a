"""Stage times of the bench frame, one frame at a time, for a given build of the
library: stage_times.py [path/to/librt_amd.so] [workload] [frames]
Prints the median and minimum over the frames of (primary, bounce levels, final
sum) in ms and the CRC32 of the last frame (two builds must print the same).  tools/ab.sh runs it alternately for two builds on the same box.
SHARE=n in the environment: rank 0's bands of an n-way split (bands of 8 rows) instead of the whole frame."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rtamd  # noqa: E402
if len(sys.argv) > 1 and sys.argv[1] != "-":
    rtamd.LIB_PATH = os.path.abspath(sys.argv[1])
import bench  # noqa: E402
name = sys.argv[2] if len(sys.argv) > 2 else "10k-spheres-4k-depth8"
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 30
sc, w, h, depth = bench.build_scene(name)
dev = torch.device("cuda", 0)
fb = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
share = int(os.environ.get("SHARE", "1"))
kw = dict(max_depth=depth, band_rows=8, band_index=0, band_count=share)
with rtamd.DeviceScene(sc) as ds:
    for _ in range(3):
        ds.render_device(fb.data_ptr(), None, w, h, **kw)
    torch.cuda.synchronize()
    ds.time_kernels(True)
    t = []
    for _ in range(frames):
        ds.render_device(fb.data_ptr(), None, w, h, **kw)
        t.append(ds.kernel_times())
import zlib  # noqa: E402
crc = zlib.crc32(fb.cpu().numpy().tobytes())
t = np.array(t)
med, mn = np.median(t, 0), t.min(0)
print(f"{os.path.basename(os.path.dirname(rtamd.LIB_PATH))}/{os.path.basename(rtamd.LIB_PATH)}: "
      f"median primary {med[0]:.3f} bounce {med[1]:.3f} sum {med[2]:.3f} = {med.sum():.3f} ms; "
      f"min {mn[0]:.3f} {mn[1]:.3f} {mn[2]:.3f} = {mn.sum():.3f} ms; frame crc {crc:08x}", flush=True)
