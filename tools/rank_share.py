#!/usr/bin/env python3
"""Times what ONE rank of an N-GPU run does per frame, on one GPU and without
any communication: the bands of rank 0 out of N (rt_frame.band_index/band_count).
Strong scaling can be no better than frame(1) / (N * frame_share(N)).
usage: rank_share.py [workload] [band_rows]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
import torch  # noqa: E402
import rtamd  # noqa: E402
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "10k-spheres-4k-depth8"
band_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc, w, h, depth = bench.build_scene(name)
dev = torch.device("cuda", 0)
fb = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
print(f"{name}, bands of {band_rows} rows", flush=True)
base = None
with rtamd.DeviceScene(sc) as ds:
    ds.set_one_launch(int(os.environ.get('ONE_LAUNCH', '1')))
    ds.set_side_stream(int(os.environ.get('SIDE', '1')))
    print(f"one launch {os.environ.get('ONE_LAUNCH', '1')}, side stream {os.environ.get('SIDE', '1')}", flush=True)
    for n in ([int(os.environ['RANK_SHARE_N'])] if 'RANK_SHARE_N' in os.environ else (1, 2, 4, 8)):
        def frame():
            ds.render_device(fb.data_ptr(), None, w, h, max_depth=depth, band_rows=band_rows,
                             band_index=0, band_count=n, stream=stream, dev_ray_count=cnt.data_ptr())
        for _ in range(3):
            frame()
        torch.cuda.synchronize()
        t = time.perf_counter()
        k = 20
        for _ in range(k):
            frame()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / k * 1e3
        base = base or ms
        ds.time_kernels(True)
        frame()
        st = ds.kernel_times()
        ds.time_kernels(False)
        print(f"N={n}: one rank's share {ms:.3f} ms per frame -> compute-only scaling efficiency "
              f"{base / (n * ms):.2f}   (primary {st[0]:.3f}, bounce levels {st[1]:.3f}, "
              f"final sum {st[2]:.3f} ms)", flush=True)
