"""Two ranks on one GPU: each renders its row bands through rt_render_device,
the bands are gathered to rank 0 (gloo through host memory here; bench.py uses
the same BandPlan with RCCL), and rank 0 must hold exactly the single-GPU frame,
at depth 1 and with bounces."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, depth, out_path):
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rtamd
    from rt_shard import BandPlan
    sc = rtamd.material_grid_scene(12, 12)
    plan = BandPlan(h, w, 16, rank, world)
    fb = plan.alloc_frame("cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    with rtamd.DeviceScene(sc) as ds:
        ds.render_device(fb.data_ptr(), None, w, h, max_depth=depth, band_rows=16,
                         band_index=rank, band_count=world,
                         stream=torch.cuda.current_stream().cuda_stream,
                         dev_ray_count=cnt.data_ptr())
        torch.cuda.synchronize()
    send, recv = plan.alloc_gather("cpu")
    plan.gather_to_rank0(fb, send, recv)
    rays = cnt.cpu()
    dist.all_reduce(rays)
    if rank == 0:
        np.save(out_path, fb[:h].cpu().numpy())
        np.save(out_path + ".rays.npy", rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("depth", [1, 5])
def test_two_ranks_rebuild_the_single_gpu_frame(tmp_path, depth):
    import torch.multiprocessing as mp
    import rtamd
    w, h = 600, 330
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), w, h, depth, out), nprocs=2, join=True)
    whole, _, n = rtamd.render(rtamd.material_grid_scene(12, 12), w, h, max_depth=depth,
                               want_ids=False)
    assert np.array_equal(np.load(out), whole)
    assert int(np.load(out + ".rays.npy")[0]) == n


def test_eight_way_split_of_the_4k_depth8_frame():
    """The multi-GPU configuration's work split, on one GPU: the 10 000-sphere
    3840x2160 depth-8 frame rendered as the eight ranks of an 8-way run would
    (bands of 8 rows, band b to rank b mod 8), one rank after the other into one
    buffer.  The result must be the whole-frame render, byte for byte, and the
    ranks' ray counts must add up to the frame's."""
    import torch
    import rtamd
    sc = rtamd.material_grid_scene(100, 100)
    w, h, depth, world = 3840, 2160, 8, 8
    dev = torch.device("cuda", 0)
    whole = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
    parts = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    with rtamd.DeviceScene(sc) as ds:
        ds.render_device(whole.data_ptr(), None, w, h, max_depth=depth, dev_ray_count=cnt.data_ptr())
        torch.cuda.synchronize()
        n_whole = int(cnt.item())
        shares = []
        for rank in range(world):
            cnt.zero_()
            ds.render_device(parts.data_ptr(), None, w, h, max_depth=depth, band_rows=8,
                             band_index=rank, band_count=world, dev_ray_count=cnt.data_ptr())
            torch.cuda.synchronize()
            assert not ds.queue_overflowed()
            shares.append(int(cnt.item()))
    assert sum(shares) == n_whole
    assert max(shares) < 1.1 * n_whole / world      # the interleaved bands balance the ranks
    assert torch.equal(whole, parts)


@pytest.mark.parametrize("depth", [1, 3])
def test_more_ranks_than_bands_and_a_ragged_last_band(depth):
    """A 200x20 frame in bands of 8 rows has three bands, the last of four rows; split
    over five ranks, two of them own nothing.  The ranks' frames together are the
    whole frame, their ray counts add up, and an empty share counts no ray."""
    import torch
    import rtamd
    sc = rtamd.material_grid_scene(12, 12)
    w, h, world = 200, 20, 5
    dev = torch.device("cuda", 0)
    whole = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
    parts = torch.zeros(h, w, 4, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    with rtamd.DeviceScene(sc) as ds:
        ds.render_device(whole.data_ptr(), None, w, h, max_depth=depth, dev_ray_count=cnt.data_ptr())
        torch.cuda.synchronize()
        n_whole = int(cnt.item())
        shares = []
        for rank in range(world):
            cnt.zero_()
            ds.render_device(parts.data_ptr(), None, w, h, max_depth=depth, band_rows=8,
                             band_index=rank, band_count=world, dev_ray_count=cnt.data_ptr())
            torch.cuda.synchronize()
            assert not ds.queue_overflowed()
            shares.append(int(cnt.item()))
    assert shares[3] == 0 and shares[4] == 0 and min(shares[:3]) > 0
    assert sum(shares) == n_whole
    assert torch.equal(whole, parts)
