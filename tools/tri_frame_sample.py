"""The triangle-heavy configuration at full size (1 M triangles + 10 000 spheres,
7680x4320, depth 8): N random pixels of the GPU frame against the CPU oracle (every
ray against every primitive), in batches so that progress shows.
usage: tri_frame_sample.py [pixels] [seed]"""
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rtamd  # noqa: E402
from oracle_py import Oracle  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
sc = rtamd.mesh_scene(1000, 500, grid=100)
w, h = 7680, 4320
fb = torch.zeros(h, w, 4, dtype=torch.uint8, device="cuda")
ids = torch.zeros(h, w, dtype=torch.int32, device="cuda")
with rtamd.DeviceScene(sc) as ds:
    ds.render_device(fb.data_ptr(), ids.data_ptr(), w, h, max_depth=8)
    torch.cuda.synchronize()
    assert not ds.queue_overflowed()
fb, ids = fb.cpu().numpy(), ids.cpu().numpy()
rng = np.random.RandomState(seed)
px = np.stack([rng.randint(0, h, n), rng.randint(0, w, n)], 1)
o = Oracle()
bad = tri = 0
t0 = time.time()
for k in range(0, n, 2048):
    p = px[k:k + 2048]
    want, want_ids, _ = o.render_pixels(sc, w, h, 8, p)
    g, gi = fb[p[:, 0], p[:, 1]], ids[p[:, 0], p[:, 1]]
    bad += int((gi != want_ids).sum()) + int((g != want).any(-1).sum())
    tri += int(((want_ids >= 0) & (want_ids < sc.t_size)).sum())
    print(f"{k + len(p)} pixels, {bad} differ, {time.time() - t0:.0f} s", flush=True)
print(f"{n} pixels of the {w}x{h} depth-8 frame (seed {seed}): {bad} differ from the oracle; "
      f"{tri} see a triangle first")
sys.exit(1 if bad else 0)
