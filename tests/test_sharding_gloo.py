"""The N > 1 path on CPU: two gloo ranks each fill their row bands (the CPU
oracle stands in for the GPU render) and gather them to rank 0, which must end
up with exactly the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, band_rows, out_path):
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import golden_scenes
    from oracle_py import Oracle
    from rt_shard import BandPlan
    sc = golden_scenes.mixed_scene(seed=71)
    plan = BandPlan(h, w, band_rows, rank, world)
    fb = plan.alloc_frame("cpu")
    o = Oracle()
    for b in plan.owned():
        r0, r1 = b * band_rows, min(h, (b + 1) * band_rows)
        part, _, _ = o.render(sc, w, h, rows=(r0, r1, 1), want_ids=False)
        fb[r0:r1] = torch.from_numpy(part[r0:r1])
    send, recv = plan.alloc_gather("cpu")
    if band_rows == 8:
        plan.gather_to_rank0(fb, send, recv)
    else:  # the overlapped form bench.py uses
        work = plan.gather_to_rank0(fb, send, recv, async_op=True)
        work.wait()
        plan.scatter_received(fb, recv)
    if rank == 0:
        np.save(out_path, fb[:h].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,band_rows,h", [(2, 8, 100), (3, 16, 90)])
def test_band_gather_rebuilds_the_frame(tmp_path, oracle, world, band_rows, h):
    import golden_scenes
    w = 160
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, band_rows, out), nprocs=world, join=True)
    whole, _, _ = oracle.render(golden_scenes.mixed_scene(seed=71), w, h, want_ids=False)
    assert np.array_equal(np.load(out), whole)


def test_band_plan_partitions_every_band_once():
    sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
    from rt_shard import BandPlan
    for h, br, world in [(2160, 8, 8), (1080, 16, 3), (7, 8, 4), (1440, 8, 1)]:
        seen = []
        for r in range(world):
            seen += BandPlan(h, 64, br, r, world).owned()
        plan = BandPlan(h, 64, br, 0, world)
        assert sorted(seen) == list(range(plan.n_bands))
        assert plan.padded_rows() >= h
        assert max(len(BandPlan(h, 64, br, r, world).owned()) for r in range(world)) == plan.max_owned
