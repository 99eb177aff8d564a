"""Where the primary pass's time goes: the bench scene's frame at depth 1 lit (primary
walk, shading, shadow walk, write) and unlit (primary walk and write only:
RT_FRAME_UNLIT), and the depth-8 frame's primary pass (the same plus the first-bounce
rays appended to the queue).  usage: primary_split.py [grid] [w] [h]"""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rtamd  # noqa: E402
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 100
w = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
h = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
sc = rtamd.material_grid_scene(grid, grid)
fb = torch.zeros(h, w, 4, dtype=torch.uint8, device="cuda")
with rtamd.DeviceScene(sc) as ds:
    for name, depth, flags in (("depth 1, lit", 1, 0), ("depth 1, unlit", 1, rtamd.RT_FRAME_UNLIT),
                               ("depth 8, primary pass", 8, 0)):
        for _ in range(3):
            ds.render_device(fb.data_ptr(), None, w, h, max_depth=depth, flags=flags)
        torch.cuda.synchronize()
        ds.time_kernels(True)
        t = []
        for _ in range(20):
            ds.render_device(fb.data_ptr(), None, w, h, max_depth=depth, flags=flags)
            t.append(ds.kernel_times()[0])
        ds.time_kernels(False)
        print(f"{name}: primary pass {np.median(t):.3f} ms (min {min(t):.3f})", flush=True)
