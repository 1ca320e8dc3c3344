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


def rank_world() -> tuple[int, int]:
    """(rank, world) of the process group, (0, 1) when torch.distributed is not in use"""
    import sys
    dist = sys.modules.get("torch.distributed")
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_frames(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous frame range [first, last) of `rank`: sizes differ by at most one, earlier ranks get the extra"""
    base, extra = divmod(total, world)
    first = rank*base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def shard_batches(first: int, last: int, batch: int) -> list[tuple[int, int]]:
    """(first, count) batches covering [first, last)"""
    return [(k, min(batch, last - k)) for k in range(first, last, batch)]


class FrameGather:
    """Two-slot asynchronous gather of equally sized byte buffers to rank 0.

    RCCL ("nccl") gathers device buffers directly. The gloo backend has no device gather, so device buffers are staged
    through host memory there (tests with several processes on one GPU, CPU-only process groups)."""

    def __init__(self, world: int, rank: int, nbytes: int, device, slots: int = 2, host_wait: bool = True):
        """`host_wait`: `wait()` returns when the gather HAS finished. RCCL's own wait only makes torch's current stream wait,
        which orders nothing for callers whose kernels and copies run on another stream (the export paths: the context's
        stream and the read-out ring's copy stream) — they could overwrite a buffer still being sent, or read one still being
        received. Callers that run everything on torch's current stream (bench.py) pass False and stay asynchronous."""
        import torch
        import torch.distributed as dist
        self.host_wait = host_wait
        self.world, self.rank, self.nbytes = world, rank, nbytes
        self.device = torch.device(device)
        self.staged = (dist.get_backend() == "gloo") and (self.device.type != "cpu")
        where = torch.device("cpu") if self.staged else self.device
        self.pending: list = [None]*slots
        self.received: list[Optional[list]] = [None]*slots
        if rank == 0:
            self.received = [[torch.empty(nbytes, dtype=torch.uint8, device=where) for _ in range(world)] for _ in range(slots)]

    def start(self, slot: int, tensor) -> None:
        import torch.distributed as dist
        self.wait(slot)
        if self.staged:
            tensor = tensor.cpu()
        self.pending[slot] = dist.gather(tensor, gather_list=self.received[slot] if self.rank == 0 else None, dst=0, async_op=True)

    def wait(self, slot: int) -> None:
        work = self.pending[slot]
        if work is not None:
            work.wait()
            if self.host_wait and not self.staged and self.device.type != "cpu":
                import torch
                torch.cuda.current_stream(self.device).synchronize()
            self.pending[slot] = None

    def wait_all(self) -> None:
        for slot in range(len(self.pending)):
            self.wait(slot)

    def frames(self, slot: int) -> list:
        """Rank 0: the `world` buffers of the last completed gather of `slot`, in rank order (on the gather's device)"""
        self.wait(slot)
        if self.staged and self.received[slot] is not None:
            return [buffer.to(self.device) for buffer in self.received[slot]]
        return self.received[slot]


def round_robin_export(world: int, rank: int, batches: list[tuple[int, int]], advance, render, emit, gather: "FrameGather | None",
                       buffers: list, frame_bytes: int) -> None:
    """Sharded export of `batches` = [(first frame, count), …] (in time order) over `world` ranks.

    Batch b is RENDERED by rank b % world; every rank ADVANCES the audio tape through every batch (the DynamicNumber
    recurrences must see all frames, and doing that redundantly costs microseconds per frame and no communication).
    After each round of `world` batches the finished frames are gathered to rank 0, which `emit`s them in frame order
    — so the encoder-side rank never holds more than one round. The gather of round r overlaps the render of round r+1
    (two alternating buffers per rank).

        advance(first, count)            compute the tape for those frames (all ranks, all batches)
        render(count, buffer)            render the tape slots [0, count) into `buffer` (the owner of the batch)
        emit(buffer, count)              rank 0: hand `count` frames at the start of `buffer` to the sink
        buffers                          two equally sized per-rank frame buffers (torch uint8 tensors)
    """
    rounds = [batches[k:k + world] for k in range(0, len(batches), world)]
    pending: list[tuple[int, list[tuple[int, int]]]] = []            # (slot, round) whose gather is in flight

    def drain(slot: int, members: list[tuple[int, int]]) -> None:
        if gather is None:
            emit(buffers[slot], members[0][1])
            return
        received = gather.frames(slot)
        if rank == 0:
            for owner, (_, count) in enumerate(members):
                emit(received[owner], count)

    for index, members in enumerate(rounds):
        slot = index % 2
        while pending and pending[0][0] == slot:                     # this buffer's previous round must have left
            drain(*pending.pop(0))
        for owner, (first, count) in enumerate(members):
            advance(first, count)
            if owner == rank:
                render(count, buffers[slot])
        if gather is not None:
            gather.start(slot, buffers[slot])                        # ranks without a batch in a short last round send stale bytes
        pending.append((slot, members))
    while pending:
        drain(*pending.pop(0))


def frame_modes(batches: list[tuple[int, int]], world: int, rank: int, warmup: Optional[int]) -> list[int]:
    """What `rank` does with every frame of a frame-loop export sharded batch-wise (batch b belongs to rank b % world):

        2  render and keep: the frame belongs to one of this rank's batches
        1  render only: within `warmup` frames before one of its batches (temporal textures need their history —
           SURVEY.md §8e; `warmup=None` means unbounded feedback: everything before an owned batch is rendered)
        0  host logic only: modules update (clocks, DynamicNumbers, video uploads), no shader is launched
    """
    total = sum(count for _, count in batches)
    modes = [0]*total
    for index, (first, count) in enumerate(batches):
        if index % world != rank:
            continue
        start = 0 if warmup is None else max(0, first - warmup)
        for k in range(start, first):
            modes[k] = max(modes[k], 1)
        for k in range(first, first + count):
            modes[k] = 2
    return modes


def sharded_frame_loop(world: int, rank: int, batches: list[tuple[int, int]], modes: list[int], step, finish_batch, emit,
                       gather: "FrameGather | None", buffers: list, frame_bytes: int) -> None:
    """Frame-loop scenes (python logic between frames, temporal textures) over `world` ranks: every rank walks ALL frames
    in order — `step(frame, mode, buffer, offset)` with the mode of `frame_modes` — and keeps the frames of its own
    batches in `buffer` at `offset`; the rounds, the gather to rank 0 and the in-order `emit` are `round_robin_export`'s.
    `finish_batch()` is called by the owner after its batch (stream synchronisation before the gather)."""
    batch_index = {first: index for index, (first, _) in enumerate(batches)}

    def advance(first: int, count: int) -> None:
        buffer = buffers[(batch_index[first]//world) % 2]
        for i in range(count):
            step(first + i, modes[first + i], buffer, i*frame_bytes)

    round_robin_export(world, rank, batches, advance, lambda count, buffer: finish_batch(), emit, gather, buffers, frame_bytes)
