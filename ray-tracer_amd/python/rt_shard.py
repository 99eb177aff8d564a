"""Row-band sharding of one frame over the ranks of a torch.distributed job.

The reference parallelises image rows with no shared state (renderer.cpp:27-31)
and ships the framebuffer back row by row (`map(from:)`, renderer.cpp:29-30).
Here bands of `band_rows` rows are dealt round-robin to the ranks (band b belongs
to rank b % world), every rank renders its bands into a frame-sized buffer of its
own, and one gather moves the bands to rank 0.  The gather is the only exchange
on the path; with the `nccl` backend it is RCCL over xGMI, with `gloo` it runs on
CPU tensors (tests).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class BandPlan:
    def __init__(self, h: int, w: int, band_rows: int, rank: int, world: int):
        if band_rows <= 0 or band_rows % 8:
            raise ValueError("band_rows must be a positive multiple of 8")
        self.h, self.w, self.band_rows = h, w, band_rows
        self.rank, self.world = rank, world
        self.n_bands = (h + band_rows - 1) // band_rows
        self.band_bytes = band_rows * w * 4
        self.max_owned = (self.n_bands + world - 1) // world

    def owned(self, rank: Optional[int] = None) -> List[int]:
        r = self.rank if rank is None else rank
        return list(range(r, self.n_bands, self.world))

    def padded_rows(self) -> int:
        """Rows of the frame buffer: whole bands, so the last band may run past h."""
        return self.n_bands * self.band_rows

    def alloc_frame(self, device) -> torch.Tensor:
        return torch.zeros((self.padded_rows(), self.w, 4), dtype=torch.uint8, device=device)

    def alloc_gather(self, device):
        send = torch.zeros((self.max_owned, self.band_bytes), dtype=torch.uint8, device=device)
        recv = None
        if self.rank == 0:
            recv = [torch.empty_like(send) for _ in range(self.world)]
        return send, recv

    def gather_to_rank0(self, fb: torch.Tensor, send: torch.Tensor, recv, async_op: bool = False):
        """Moves every rank's bands into rank 0's `fb`.  No-op for world == 1.
        With async_op the collective's work handle is returned and rank 0 must
        call scatter_received() after waiting on it."""
        if self.world == 1:
            return None
        bands = fb.view(self.n_bands, self.band_bytes)
        mine = bands[self.rank::self.world]
        send[:mine.shape[0]].copy_(mine)   # device -> host when send lives on the CPU (gloo)
        work = dist.gather(send, recv, dst=0, async_op=async_op)
        if async_op:
            return work
        self.scatter_received(fb, recv)
        return None

    def scatter_received(self, fb: torch.Tensor, recv) -> None:
        """Rank 0: places the gathered bands of the other ranks into `fb`."""
        if self.world == 1 or self.rank != 0:
            return
        bands = fb.view(self.n_bands, self.band_bytes)
        for r in range(1, self.world):
            k = len(self.owned(r))
            bands[r::self.world].copy_(recv[r][:k])
