#!/usr/bin/env python3
"""bench.py — Mrays/s of the trace/shade path on N MI355X of one node.

One "step" = one frame of the workload rendered through the C ABI
(rt_render_device) with the scene already resident in HBM.  With N > 1 the
frame's row bands are dealt round-robin to the ranks (one process per GPU) and
the bands are gathered to rank 0 over RCCL inside the timed region; total work
is fixed, so scaling is "strong".

Prints ONE JSON line on rank 0.  Ray counts come from the device-side counter.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))

import rtamd  # noqa: E402
from rt_shard import BandPlan  # noqa: E402

PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X vector FP32, FMA counted as 2 flop
# The path may not fuse (results must equal the reference's unfused arithmetic),
# so a lane does at most one flop per vector instruction: half of the peak above.
PEAK_UNFUSED_TLANEOPS = PEAK_FP32_VALU_TFLOPS / 2
STAGES = ("primary", "bounce", "compose")   # rt_scene_kernel_times order
PEAK_HBM_GBS = 8000.0           # MI355X HBM3E
FLOP_PER_SPHERE_TEST = 25       # maths.cpp:126-133 up to the discriminant test
FLOP_PER_TRIANGLE_TEST = 107    # maths.cpp:68-116 full path

WORKLOADS = {
    # name: (grid rows, grid cols, width, height, depth, materials, mesh quads or None)
    "10k-spheres-4k-depth8": (100, 100, 3840, 2160, 8, True, None),
    "10k-spheres-4k-depth1": (100, 100, 3840, 2160, 1, False, None),
    "10k-spheres-8k-depth8": (100, 100, 7680, 4320, 8, True, None),
    "1k-spheres-1080p-depth4": (32, 32, 1920, 1080, 4, True, None),
    "1k-spheres-1080p-depth1": (32, 32, 1920, 1080, 1, False, None),
    "shipped-1440p-depth1": (10, 10, 2560, 1440, 1, False, None),
    # the triangle-heavy configuration: 1 M-triangle height field + the 10k spheres
    "1m-triangles-8k-depth8": (100, 100, 7680, 4320, 8, True, (1000, 500)),
    "1m-triangles-4k-depth8": (100, 100, 3840, 2160, 8, True, (1000, 500)),
}
PLAIN_LOOP_MAX_PRIMS = 20000   # beyond this the every-primitive loop takes minutes per frame


def build_scene(name):
    row, col, w, h, depth, mats, mesh = WORKLOADS[name]
    if mesh:
        sc = rtamd.mesh_scene(mesh[0], mesh[1], grid=row, materials=mats)
    else:
        sc = rtamd.material_grid_scene(row, col) if mats else rtamd.grid_scene(row, col)
    return sc, w, h, depth


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sc, w, h, depth, budget_s=24.0):
    """Times the CPU oracle (oracle/, the bit-exact restatement of the reference
    renderer) on a bounded sample of the same frame: one warm-up run, then three
    timed runs of the same sample, the median reported.  The sample is sized so
    that the four runs take about budget_s together."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from oracle_py import Oracle
    o = Oracle()
    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    per_run = budget_s / 4.0
    build = "oracle/rt_oracle.c, gcc -O2 -fopenmp -ffp-contract=off, schedule(dynamic) over rows"
    if sc.t_size > PLAIN_LOOP_MAX_PRIMS:
        # a single row of this frame takes the CPU minutes: sample pixels instead
        rng = np.random.RandomState(2024)
        def pixels(n):
            return np.stack([rng.randint(0, h, n), rng.randint(0, w, n)], 1)
        def run(px):
            t = time.perf_counter()
            _, _, rays = o.render_pixels(sc, w, h, depth, px)
            return rays, time.perf_counter() - t
        _, dt0 = run(pixels(4 * threads))          # calibration
        n = int(max(4 * threads, min(200000, 4 * threads * per_run / max(dt0, 1e-3))))
        px = pixels(n)
        run(px)                                    # warm-up
        runs = [run(px) for _ in range(3)]
        rays = runs[0][0]
        times = sorted(dt for _, dt in runs)
        what = f"{n} random pixels ({rays} rays) of the same {w}x{h} frame"
    else:
        # calibrate on a sparse sample, then size the real one
        step0 = max(1, h // 32)
        t = time.perf_counter()
        o.render(sc, w, h, depth, rows=(step0 // 2, h, step0), want_ids=False)
        dt0 = time.perf_counter() - t
        rows0 = len(range(step0 // 2, h, step0))
        want_rows = int(min(h, max(8, per_run / (dt0 / rows0))))
        step = max(1, h // want_rows)
        sel = (step // 2, h, step)
        def run():
            t = time.perf_counter()
            _, _, n = o.render(sc, w, h, depth, rows=sel, want_ids=False)
            return n, time.perf_counter() - t
        run()                                      # warm-up
        runs = [run() for _ in range(3)]
        rays = runs[0][0]
        times = sorted(dt for _, dt in runs)
        what = (f"every {step}th row ({len(range(*sel))} of {h} rows, {rays} rays) of the same "
                f"{w}x{h} frame")
    dt = times[1]
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "cpu_model": cpu_model(),
        "runs_s": [round(x, 3) for x in times], "protocol": "1 warm-up + 3 timed runs, median",
        "sample": f"{what}; {build}; median {dt:.2f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--band-rows", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plain-loop", action="store_true",
                    help="skip the every-ray-every-primitive leg (profiling runs)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend for --gpus > 1; gloo stages the bands through "
                         "host memory and exists to rehearse the multi-rank path on one GPU")
    ap.add_argument("--screen", choices=["auto", "off", "beam", "bvh"], default="auto",
                    help="work avoidance mode (rt_set_screen); off = plain every-ray-"
                         "every-sphere loop")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="frames rendered concurrently for `value`, each on its own stream with "
                         "its own resident scene copy (default 1 = strictly one frame at a time, "
                         "for every N; with --gpus N > 1 a second leg with four in flight is "
                         "measured as well and reported beside `value`)")
    ap.add_argument("--no-in-flight-leg", action="store_true",
                    help="--gpus N > 1: skip the second leg with four frames in flight")
    ap.add_argument("--one-launch", type=int, default=1, choices=[0, 1],
                    help="bounce levels in one launch (default) or a launch per stage and level")
    ap.add_argument("--side-stream", type=int, default=None, choices=[0, 1],
                    help="shadow walks of a level on a side stream (default: on with one frame "
                         "in flight, off with several)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    n_dev = torch.cuda.device_count()
    local_rank = local_rank % max(1, n_dev)   # a rehearsal may put several ranks on one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")

    name = args.workload or default_workload()
    sc, w, h, depth = build_scene(name)
    n_tests_per_ray = FLOP_PER_SPHERE_TEST * sc.s_size + FLOP_PER_TRIANGLE_TEST * sc.t_size

    band_rows = args.band_rows
    plan = BandPlan(h, w, band_rows, rank, world)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    rtamd.set_screen(args.screen)   # the mode scenes created below start with
    # `value` is measured with ONE frame in flight whatever N is: with N > 1 that is
    # the strong scaling of a frame, which is what "scaling": "strong" says.  The
    # bounce levels of a frame are one launch whose waves work through chains of up
    # to max_depth dependent levels, so a rank's share of a frame cannot keep the
    # chip busy by itself; a renderer of image sequences hides that by keeping
    # several frames in flight (each on its own stream, with its own resident scene
    # copy and frame buffer), and with N > 1 a second leg measures that too and
    # reports it beside `value` ("frames_in_flight_4"), never instead of it.
    n_buf = max(1, args.frames_in_flight)

    class Slot:
        pass

    def make_slots(n, side):
        out = []
        for _ in range(n):
            sl = Slot()
            sl.scene = rtamd.DeviceScene(sc)
            sl.scene.set_side_stream(side)
            sl.scene.set_one_launch(args.one_launch)
            sl.stream = torch.cuda.Stream(device=dev)
            sl.fb = plan.alloc_frame(dev)
            sl.send, sl.recv = plan.alloc_gather(comm_dev)
            sl.pending = None
            out.append(sl)
        return out

    def timed_leg(slots):
        """W warm-up frames, then K timed frames between barriers; returns wall
        seconds (max over ranks), rays of all ranks, this rank's rays per frame and
        the mean first-to-last-launch device time of a frame."""
        n = len(slots)
        events = []

        def finish(sl):
            # wait for the slot's gather and, on rank 0, put the other ranks' bands
            # in place: the frame is then complete on GPU 0
            if sl.pending is not None:
                sl.pending.wait()
                sl.pending = None
                plan.scatter_received(sl.fb, sl.recv)

        def step(marked, k=0):
            sl = slots[k % n]
            with torch.cuda.stream(sl.stream):
                finish(sl)
                if marked:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                sl.scene.render_device(sl.fb.data_ptr(), None, w, h, max_depth=depth,
                                       band_rows=band_rows, band_index=rank, band_count=world,
                                       stream=sl.stream.cuda_stream, dev_ray_count=cnt.data_ptr())
                if marked:
                    e1.record()
                    events.append((e0, e1))
                sl.pending = plan.gather_to_rank0(sl.fb, sl.send, sl.recv, async_op=True)

        def drain():
            for sl in slots:
                with torch.cuda.stream(sl.stream):
                    finish(sl)

        for k in range(max(args.warmup, n)):
            step(False, k)
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        cnt.zero_()
        torch.cuda.synchronize()
        # the timed frames carry no event records: a record between two launches is a
        # packet of its own that the next launch waits for (0.03 ms per 0.9 ms frame)
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(False, k)
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rays_timed = cnt.clone()
        # a frame's own first-to-last-launch time: a few more frames, with marks
        for k in range(min(args.steps, 8)):
            step(True, k)
        drain()
        torch.cuda.synchronize()
        for sl in slots:
            if sl.scene.queue_overflowed():
                raise SystemExit("bounce-ray queue overflowed: the frames are incomplete, "
                                 "no number reported")
        rays = rays_timed.clone()
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(rays, op=dist.ReduceOp.SUM)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # with several frames in flight the first-to-last-launch intervals overlap:
        # a frame's latency, not 1/throughput
        return {"dt": float(tmax.item()), "total_rays": int(rays.item()),
                "my_rays_per_frame": int(rays_timed.item()) / args.steps,
                "frame_latency_ms": float(np.mean([e0.elapsed_time(e1) for e0, e1 in events]))}

    # several frames in flight: one stream per frame, no side streams inside a
    # frame (the device has 4 hardware queues; streams sharing one serialise)
    side = args.side_stream if args.side_stream is not None else (1 if n_buf == 1 else 0)
    slots = make_slots(n_buf, side)
    leg = timed_leg(slots)
    dt, total_rays = leg["dt"], leg["total_rays"]
    my_rays_per_frame = leg["my_rays_per_frame"]

    # ---- one frame at a time, stage by stage (not part of `value`) ----
    # HIP events recorded by the library on the frame's own stream around the
    # primary pass (k_render), the bounce levels (k_trace, or the chain of
    # k_walk / k_shade launches) and the final sum (k_compose).
    def staged_frames(n):
        sl = slots[0]
        sl.scene.time_kernels(True)
        per = []
        for _ in range(n):
            with torch.cuda.stream(sl.stream):
                sl.scene.render_device(sl.fb.data_ptr(), None, w, h, max_depth=depth,
                                       band_rows=band_rows, band_index=rank, band_count=world,
                                       stream=sl.stream.cuda_stream, dev_ray_count=cnt.data_ptr())
            per.append(sl.scene.kernel_times())   # waits for the frame
        sl.scene.time_kernels(False)
        return [float(np.mean([p[k] for p in per])) for k in range(3)]

    staged_frames(2)
    stage_ms = staged_frames(min(10, max(2, args.steps)))
    one_frame_ms = float(sum(stage_ms))

    # ---- the same frames with a camera that moves every frame (not part of `value`) ----
    # The spheres' lists by screen cell and the eye-relative records belong to one
    # view; the timed frames above share theirs, like any still camera.  Here they
    # are made anew for every frame.
    def moving_camera_frames(n):
        sl = slots[0]
        cams = [rtamd.default_camera(), rtamd.default_camera()]
        cams[1].origin[0] += 1e-3
        with torch.cuda.stream(sl.stream):
            for k in range(2):
                sl.scene.render_device(sl.fb.data_ptr(), None, w, h, max_depth=depth, cam=cams[k],
                                       band_rows=band_rows, band_index=rank, band_count=world,
                                       stream=sl.stream.cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(n):
                sl.scene.render_device(sl.fb.data_ptr(), None, w, h, max_depth=depth,
                                       cam=cams[k % 2], band_rows=band_rows, band_index=rank,
                                       band_count=world, stream=sl.stream.cuda_stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            # back to the bench's own view
            sl.scene.render_device(sl.fb.data_ptr(), None, w, h, max_depth=depth,
                                   band_rows=band_rows, band_index=rank, band_count=world,
                                   stream=sl.stream.cuda_stream)
            torch.cuda.synchronize()
        return ms

    moving_ms = moving_camera_frames(max(4, min(20, args.steps)))

    # ---- N > 1: the same split with four frames in flight, beside `value` ----
    in_flight = None
    if world > 1 and n_buf == 1 and not args.no_in_flight_leg:
        more = make_slots(4, 0)
        leg4 = timed_leg(more)
        in_flight = {"frames_in_flight": 4, "ms_per_frame": leg4["dt"] / args.steps * 1e3,
                     "mrays_per_s": leg4["total_rays"] / leg4["dt"] / 1e6,
                     "frame_latency_ms": leg4["frame_latency_ms"],
                     "note": "throughput of a sequence of frames, four in flight per rank; "
                             "not the strong scaling of one frame, which is `value`"}
        for sl in more:
            sl.scene.close()

    if rank == 0:
        value = total_rays / dt / 1e6
        out = {
            "metric": "Mrays/sec (primary+secondary)", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": name, "width": w, "height": h, "max_depth": depth,
                       "spheres": sc.s_size, "triangles": sc.t_size, "lights": sc.l_size,
                       "materials": sc.mat_sphere is not None,
                       "rays_per_frame": total_rays // args.steps,
                       # device-counted, i.e. the rays actually traced: a bounce hit casts
                       # no shadow ray towards a light whose contribution truncates to zero
                       # in every channel (same image, fewer rays than counting one shadow
                       # ray per light and hit would give)
                       "rays_counted": "traced",
                       "sharding": f"row bands of {band_rows}, round-robin over {world} GPU(s)",
                       "screen": args.screen, "frames_in_flight": n_buf,
                       "one_launch": args.one_launch},
            "frame_latency_ms": leg["frame_latency_ms"],
            # this rank's share of the frame with nothing else on its GPU and no
            # gather: sum of the three stages
            "one_frame_ms": one_frame_ms,
            "one_frame_mrays_per_s": my_rays_per_frame / (one_frame_ms * 1e-3) / 1e6,
            # the records that belong to a view (eye-relative spheres, sphere lists by
            # screen cell) are kept while the camera stands still, as in the timed frames
            "view_records": {"kept_while_the_camera_stands_still": True,
                             "ms_per_frame_with_a_camera_that_moves_every_frame": moving_ms},
        }
        if in_flight is not None:
            out["frames_in_flight_4"] = in_flight
        # framebuffer bytes written per stage: this rank's rows x 4 bytes by the primary
        # pass and again by the final sum (which only runs when the frame has bounces)
        owned_rows = sum(min(band_rows, h - r0) for r0 in range(rank * band_rows, h, world * band_rows))
        fb_px = owned_rows * w * 4
        fb_bytes = [fb_px, 0, fb_px if stage_ms[2] > 0 else 0]
        out["roofline"] = roofline(name, args, world, stage_ms, my_rays_per_frame, n_tests_per_ray,
                                   fb_bytes)
        if world == 1 and args.screen != "off" and not args.no_plain_loop:
            if sc.t_size + sc.s_size > PLAIN_LOOP_MAX_PRIMS:
                out["roofline"]["plain_loop"] = None   # minutes per frame at this size: not run
            else:
                # the same frame with every ray tested against every primitive: the
                # formulation whose FLOP count the FP32 roofline is defined on
                sl = slots[0]
                sl.scene.set_screen("off")
                staged_frames(1)
                bf_ms = float(sum(staged_frames(2)))
                sl.scene.set_screen(args.screen)
                bf = my_rays_per_frame * n_tests_per_ray / (bf_ms * 1e-3) / 1e12
                out["roofline"]["plain_loop"] = {
                    "kernel_ms": bf_ms, "achieved": bf, "peak": PEAK_FP32_VALU_TFLOPS,
                    "unit": "TFLOP/s", "frac": bf / PEAK_FP32_VALU_TFLOPS,
                    "flop_per_ray": n_tests_per_ray,
                    "mrays_per_s": my_rays_per_frame / (bf_ms * 1e-3) / 1e6,
                    "note": "algorithmic FLOP = rays x (25 x spheres + 107 x triangles) over the "
                            "time of the frame rendered by the plain loops (rt_scene_set_screen 0)"}
                # how much of that arithmetic the shipped path proves unnecessary
                out["roofline"]["work_avoidance_factor"] = bf_ms / one_frame_ms
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, w, h, depth)
        print(json.dumps(out))
    for sl in slots:
        sl.scene.close()
    if world > 1:
        dist.destroy_process_group()


def kernel_source_sha():
    """Hash of the device code's source files: stored beside the profiled counters,
    so that instruction counts of one build are never combined with stage times of
    another."""
    import hashlib
    hsh = hashlib.sha256()
    d = os.path.join(REPO, "ray-tracer_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, fn), "rb") as f:
                hsh.update(fn.encode() + b"\0" + f.read())
    return hsh.hexdigest()[:16]


def profiled_counters(workload):
    """Per-frame instruction and traffic counts of the library's kernels, by
    stage, from rocprofv3 --pmc passes of this same program with one frame in
    flight (tools/profile.sh writes them; instruction counts do not depend on
    timing).  None when that workload has not been profiled."""
    try:
        with open(os.path.join(REPO, "profiles", "counters.json")) as f:
            return json.load(f).get(workload)
    except (OSError, ValueError):
        return None


def roofline(name, args, world, stage_ms, rays_per_frame, flop_per_ray, fb_bytes=None):
    """Where the kernels of one frame sit against the vector-FP32 ceiling.

    achieved = executed vector lane-operations per second of the stage that takes
    longest: SQ_INSTS_VALU (wave instructions per launch, from the committed
    counter passes) x active lanes per instruction / the stage's duration
    measured here with HIP events.  A lane-operation is at most one flop because
    this path may not fuse, so the ceiling is half the FMA peak.  The fraction
    cannot exceed 1 whatever work the hierarchies avoid."""
    prof = profiled_counters(name) if (world == 1 and args.screen == "auto") else None
    stale = None
    if prof is not None and prof.get("source_sha") != kernel_source_sha():
        # the counters were taken from another build of the kernels: their
        # instruction counts say nothing about this one's
        stale = prof.get("source_sha")
        prof = None
    stages = {}
    for k, st in enumerate(STAGES):
        e = {"ms": stage_ms[k]}
        c = (prof or {}).get("stages", {}).get(st)
        if c and stage_ms[k] > 0 and (prof or {}).get("one_launch", 1) == args.one_launch:
            lane_ops = c["valu_wave_insts"] * c["lanes_per_valu_inst"]
            e.update({
                "kernels": c["kernels"], "ms_rocprof": c["ms"],
                "valu_wave_insts": c["valu_wave_insts"],
                "lanes_per_valu_inst": c["lanes_per_valu_inst"],
                "tlaneops_per_s": lane_ops / (stage_ms[k] * 1e-3) / 1e12,
                "frac": lane_ops / (stage_ms[k] * 1e-3) / 1e12 / PEAK_UNFUSED_TLANEOPS,
                # wave instructions x 2 cycles over the SIMD-cycles of the launch
                "valu_issue_frac": c["valu_wave_insts"] * 2 /
                                   (1024 * c["clock_ghz"] * 1e9 * stage_ms[k] * 1e-3),
                "hbm_bytes": c["hbm_bytes"],
                "hbm_gbs": c["hbm_bytes"] / (stage_ms[k] * 1e-3) / 1e9,
                # of the wave-cycles of the stage: parked at a wait (memory, LDS), and
                # the mean number of waves resident per SIMD
                "waves_parked_frac": c.get("wait_any_frac"),
                "occupancy_waves_per_simd": c.get("occupancy_waves_per_simd"),
                # cycles the LDS spent on bank conflicts over its active cycles
                # (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE): the walks' stacks are laid
                # out word (row x 256 + thread) so that this is zero
                "lds_bank_conflict_frac": c.get("lds_bank_conflict_frac"),
            })
        if fb_bytes and fb_bytes[k] and stage_ms[k] > 0:
            # the framebuffer bytes the stage writes (4 per pixel: the primary pass
            # writes every pixel, the final sum rewrites them), as a rate
            e["fb_write_bytes"] = fb_bytes[k]
            e["fb_write_gbs"] = fb_bytes[k] / (stage_ms[k] * 1e-3) / 1e9
        stages[st] = e
    dom = max(STAGES, key=lambda st: stages[st]["ms"])
    d = stages[dom]
    r = {"bound": "valu_fp32", "kernel": dom, "kernel_ms": d["ms"],
         "achieved": d.get("tlaneops_per_s"), "peak": PEAK_UNFUSED_TLANEOPS, "unit": "TFLOP/s",
         "achieved_counts": "executed vector lane-operations (moves, integer and compare "
                            "instructions included): an upper bound on the flop rate",
         "frac": d.get("frac"), "traffic": d.get("hbm_bytes"),
         "stages": stages,
         "note": "achieved = vector lane-operations executed per second by the longest stage "
                 "(wave instructions x active lanes, rocprofv3 counters in profiles/counters.json) "
                 "over its duration measured here with HIP events, one frame on the GPU; peak = "
                 "157.3 / 2 because results must equal unfused arithmetic (1 flop per lane and "
                 "instruction at most); not every vector instruction is a flop, so this bounds "
                 "the flop rate from above"}
    if d.get("hbm_bytes") is not None:
        r["hbm"] = {"achieved": d["hbm_gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": d["hbm_gbs"] / PEAK_HBM_GBS}
    if stale is not None:
        r["note"] += (f"; profiles/counters.json was taken from another build of the kernels "
                      f"(source hash {stale}, this tree {kernel_source_sha()}): achieved is null "
                      f"until tools/profile.sh is run again")
    elif prof is None:
        r["note"] += "; no counter passes committed for this workload/mode: achieved is null"
    return r


def default_workload():
    return "10k-spheres-4k-depth8"


if __name__ == "__main__":
    main()
