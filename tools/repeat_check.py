"""The same frame rendered several times, with a launch per bounce level and with
one launch for all levels: every frame must be byte-identical (scheduling inside the
kernels differs from run to run, results must not).
usage: repeat_check.py [workload] [frames]"""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rtamd  # noqa: E402
import bench  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "1k-spheres-1080p-depth4"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc, w, h, depth = bench.build_scene(name)
dev = torch.device("cuda", 0)
ref = None
bad = 0
for one in (0, 1):
    with rtamd.DeviceScene(sc) as ds:
        ds.set_one_launch(one)
        for k in range(frames):
            fb = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
            ds.render_device(fb.data_ptr(), None, w, h, max_depth=depth)
            torch.cuda.synchronize()
            if ds.queue_overflowed():
                raise SystemExit("queue overflowed")
            a = fb.cpu().numpy()
            if ref is None:
                ref = a
            n = int((a != ref).any(-1).sum())
            bad += n
            if n:
                print(f"one_launch {one} frame {k}: {n} pixels differ from the first frame", flush=True)
print(f"{name}: {2 * frames} frames, {bad} differing pixels in all")
sys.exit(1 if bad else 0)
