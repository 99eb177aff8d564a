#!/usr/bin/env python3
"""One rank's share of an N-GPU frame (bands of rank 0 of N) with F frames in
flight: F resident copies of the scene, each frame on its own stream.  Prints the
time per frame; the latency floor of a frame's chain of dependent kernels hides
behind the other frames' work when the chip is not saturated by one frame.
usage: frames_in_flight.py [workload]"""
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
sc, w, h, depth = bench.build_scene(name)
dev = torch.device("cuda", 0)
SIDE = int(os.environ.get('SIDE', '0'))
ONE = int(os.environ.get('ONE_LAUNCH', '1'))
print(f'one launch {ONE}, side stream {SIDE}', flush=True)
flights = [int(x) for x in os.environ.get('FLIGHTS', '1,2,3,4').split(',')]
for n in [int(x) for x in os.environ.get('SHARES', '1,2,4,8').split(',')]:
    base = None
    for flight in flights:
        scenes = [rtamd.DeviceScene(sc) for _ in range(flight)]
        for d in scenes:
            d.set_side_stream(SIDE)
            d.set_one_launch(ONE)
        streams = [torch.cuda.Stream(device=dev) for _ in range(flight)]
        fbs = [torch.zeros(h, w, 4, dtype=torch.uint8, device=dev) for _ in range(flight)]
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()

        def frame(k):
            s = k % flight
            scenes[s].render_device(fbs[s].data_ptr(), None, w, h, max_depth=depth, band_rows=8,
                                    band_index=0, band_count=n, stream=streams[s].cuda_stream,
                                    dev_ray_count=cnt.data_ptr())
        for k in range(2 * flight):
            frame(k)
        torch.cuda.synchronize()
        t = time.perf_counter()
        K = 40
        for k in range(K):
            frame(k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / K * 1e3
        base = base or ms
        print(f"share 1/{n}, {flight} frame(s) in flight: {ms:.3f} ms per frame ({base / ms:.2f}x)", flush=True)
        for d in scenes:
            d.close()
