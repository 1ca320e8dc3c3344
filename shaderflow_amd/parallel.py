"""
Multi-GPU offline export: contiguous frame ranges per rank + gather of finished frames to the encoding rank.

The reference is single-process; this is the sharded export of the north star. One process per GPU
(`torch.distributed`, backend "nccl" = RCCL on ROCm; "gloo" on CPU for the tests). Frames are independent once the
audio tape is known, except for the DynamicNumber recurrences, which every rank replays from frame 0 on its own
device (a few kernels over ~1 KB per frame: far cheaper than communicating state and bit-identical by
construction). The only exchange step is the gather of finished RGB8 frames to rank 0, which owns the encoder
pipe: `dist.gather` is a group of point-to-point sends, so on MI355X's fully connected xGMI every peer streams
over its own link to rank 0 (no ring, nothing to bucket). The gather of batch i is asynchronous and overlaps the
render of batch i+1 (two alternating frame buffers per rank).
"""
from __future__ import annotations

from typing import Optional


def shard_frames(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous frame range [first, last) of `rank`: sizes differ by at most one, earlier ranks get the extra"""
    base, extra = divmod(total, world)
    first = rank*base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def shard_batches(first: int, last: int, batch: int) -> list[tuple[int, int]]:
    """(first, count) batches covering [first, last)"""
    return [(k, min(batch, last - k)) for k in range(first, last, batch)]


class FrameGather:
    """Two-slot asynchronous gather of equally sized byte buffers to rank 0"""

    def __init__(self, world: int, rank: int, nbytes: int, device, slots: int = 2):
        import torch
        self.world, self.rank, self.nbytes = world, rank, nbytes
        self.pending: list = [None]*slots
        self.received: list[Optional[list]] = [None]*slots
        if rank == 0:
            self.received = [[torch.empty(nbytes, dtype=torch.uint8, device=device) for _ in range(world)] for _ in range(slots)]

    def start(self, slot: int, tensor) -> None:
        import torch.distributed as dist
        self.wait(slot)
        self.pending[slot] = dist.gather(tensor, gather_list=self.received[slot] if self.rank == 0 else None, dst=0, async_op=True)

    def wait(self, slot: int) -> None:
        work = self.pending[slot]
        if work is not None:
            work.wait()
            self.pending[slot] = None

    def wait_all(self) -> None:
        for slot in range(len(self.pending)):
            self.wait(slot)

    def frames(self, slot: int) -> list:
        """Rank 0: the `world` buffers of the last completed gather of `slot`, in rank order"""
        self.wait(slot)
        return self.received[slot]
