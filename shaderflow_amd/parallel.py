"""
Multi-GPU offline export (one process per GPU; the reference is single-process).

Frames are independent once the audio tape is known, except for the DynamicNumber recurrences, which every rank replays on
its own device (a few kernels over ~1 KB per frame: far cheaper than communicating state, bit-identical by construction).
What is left to decide is how the finished RGB8 frames reach the ONE process that owns the sink (rank 0). Two modes
(`SHADERFLOW_SHARD`, default "host"):

* **device** — the north star's design: every rank renders ONE CONTIGUOUS frame range (`shard_frames`) into HBM — a whole
  60 s 4K export is 89.6 GB of RGB8, an eighth of it 11.2 GB: resident, not streamed — and sends it chunk by chunk to rank 0
  (`RangeTransfer`: grouped point-to-point sends = the RCCL gather; on the fully connected xGMI fabric every peer has its own
  link to rank 0 — 76 GB/s per direction in the KFD topology of the pool's nodes (DESIGN.md §6 has the per-link budget: RCCL's
  point-to-point kernels are assumed at ≈ 75 % of that, SDMA peer copies at ≈ 88 %) against 24.9 MB x frames/s per rank, or 12.4 MB as
  yuv420p: nothing to bucket). The frames then sit in rank 0's HBM;
  to reach a HOST sink they all cross rank 0's single PCIe link: ceiling ≈ 55 GB/s / 24.9 MB ≈ 2 200 frames/s at 4K whatever N is.
* **device-sdma** — the same layout as "device", but the gather is not a collective: rank 0 exports its resident buffer as an IPC
  handle, every other rank maps it and copies its finished chunks to where they belong on SDMA engines NAMED through HSA
  (`SdmaTransfer`, sfx_peer_*: a thread of the context issues hsa_amd_memory_async_copy_on_engine on the two engines HSA recommends
  for the pair of GPUs, up to four copies in flight): the engines move the bytes over the rank's own xGMI link at close to link rate
  and no compute unit is taken from the render — what RCCL's point-to-point kernels cannot offer. A gloo side channel carries
  "chunk k has landed".
* every mode moves SINK frames: rgb24, or — `scene.main(pixel_format="yuv420p")` — planar frames converted on the rank that rendered
  them (half the bytes per link; round 5).
* **host** — every rank reads its finished frames out over ITS OWN PCIe link into a shared-memory ring and rank 0's native
  writer thread interleaves them in frame order (`HostDelivery`, csrc/shm_ring.inc): no collective on the data path, ceiling
  N x min(render, PCIe) until the sink or host memory bandwidth binds. Here batches alternate between the ranks (batch b on
  rank b % N): the sink consumes in frame order, so with contiguous ranges only one rank's link would be busy at a time unless
  whole ranges were buffered on the host; alternating batches keep every link busy with two batches of host ring per rank.

torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the tests) provides rendezvous, barriers and the device-mode sends.
"""
from __future__ import annotations

import os
from typing import Optional


def shard_mode() -> str:
    """How a sharded export delivers frames to the sink's process: "host" (per-rank PCIe + shared memory) or "device" (RCCL)"""
    mode = os.environ.get("SHADERFLOW_SHARD", "host").strip().lower()
    if mode not in ("host", "device", "device-sdma"):
        raise ValueError(f"SHADERFLOW_SHARD={mode!r}: expected 'host', 'device' or 'device-sdma'")
    return mode


def rank_world() -> tuple[int, int]:
    """(rank, world) of the process group, (0, 1) when torch.distributed is not in use"""
    import sys
    dist = sys.modules.get("torch.distributed")
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def is_sharded() -> bool:
    """An export takes the multi-rank code paths: a process group with more than one rank — or, with SHADERFLOW_FORCE_DIST=1, a group
    of ONE rank (the RCCL backend, its streams and the shard modes exercised end to end on a single GPU: tests/test_gpu_rccl.py)"""
    rank, world = rank_world()
    if world > 1:
        return True
    import sys
    dist = sys.modules.get("torch.distributed")
    return bool(os.environ.get("SHADERFLOW_FORCE_DIST") == "1" and dist is not None and dist.is_available() and dist.is_initialized())


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


# ---- host mode: per-rank read-out into shared memory -------------------------------------------------------------------------

def shm_slots_that_fit(frame_bytes: int, slots: int, world: int, directory: str = "/dev/shm") -> int:
    """The ring of `world` ranks x `slots` frames lives in a tmpfs segment: writing past what the filesystem can back raises SIGBUS
    (a container's /dev/shm may be 64 MB), so the ring is cut to half of the free space — or refused, loudly, when not even two
    frames per rank fit — before anything is mapped"""
    try:
        stat = os.statvfs(directory)
    except OSError:
        return slots
    fit = (stat.f_bavail*stat.f_frsize//2)//max(1, frame_bytes*world)
    if fit < 2:
        raise RuntimeError(f"{directory} has {stat.f_bavail*stat.f_frsize >> 20} MiB free: the cross-process frame ring needs at least "
                           f"{(2*frame_bytes*world*2) >> 20} MiB for {world} ranks; use SHADERFLOW_SHARD=device or enlarge it")
    return int(min(slots, fit))


class HostDelivery:
    """This rank's end of the cross-process frame queue (csrc/shm_ring.inc): `push` frames in the order this rank finishes them;
    rank 0 also starts the writer, which hands the frames of all ranks to `fileno` in the order `runs` = [(rank, count), …]."""

    def __init__(self, context, world: int, rank: int, frame_bytes: int, slots: int, fileno: Optional[int], runs: list[tuple[int, int]]):
        import ctypes as C

        import torch.distributed as dist

        from shaderflow_amd import _native as N
        self.N, self.C = N, C
        self.rank, self.world, self.frame_bytes, self.pushed, self.failed = rank, world, frame_bytes, 0, False
        # one segment name, one ring size — and ONE verdict on whether /dev/shm can hold it: rank 0's, for the whole group (a rank that
        # raised by itself here would leave the others in the broadcast)
        name = [f"/shaderflow-{os.getpid()}-{id(self) & 0xffffff:x}", 0, None]
        try:
            name[1] = shm_slots_that_fit(frame_bytes, slots, world)
        except RuntimeError as error:
            name[2] = str(error)
        if world > 1:
            dist.broadcast_object_list(name, src=0)
        if name[2]:
            raise RuntimeError(name[2])
        slots = int(name[1])
        self.handle = N.Handle()
        failure = None
        try:
            N.check(N.lib().sfx_shm_create(context.handle, name[0].encode(), rank, world, frame_bytes, slots, C.byref(self.handle)))
        except Exception as error:                                  # tell the group instead of leaving it in a barrier
            failure = f"rank {rank}: {error}"
        if world > 1:
            reports = [None]*world
            dist.all_gather_object(reports, failure)                # every rank has mapped the segment … or everybody learns who has not
            failure = next((report for report in reports if report), None)
        if failure:
            if self.handle.value:
                N.lib().sfx_shm_destroy(self.handle)
            raise RuntimeError(f"cross-process frame ring: {failure}")
        if rank == 0:
            N.check(N.lib().sfx_shm_unlink(self.handle))            # … so its name can go: a crash leaves nothing in /dev/shm
            # without a sink (fileno None: a benchmark or a freewheeling run under torchrun) the writer still consumes every frame,
            # in the same order, and discards it — producers would otherwise fill their rings and wait for a consumer that never comes
            ranks = (C.c_int32*max(1, len(runs)))(*[r for r, _ in runs])
            counts = (C.c_int32*max(1, len(runs)))(*[c for _, c in runs])
            N.check(N.lib().sfx_shm_drain(self.handle, -1 if fileno is None else fileno, ranks, counts, len(runs)))

    def push(self, pointer: int, count: int) -> None:
        for i in range(count):
            self.N.check(self.N.lib().sfx_shm_push(self.handle, self.C.c_void_p(pointer + i*self.frame_bytes)))
        self.pushed += count

    def wait(self, frames: int) -> None:
        """The first `frames` frames pushed by this rank have left their device buffers"""
        self.N.check(self.N.lib().sfx_shm_wait(self.handle, frames))

    def abort(self) -> None:
        """This rank cannot go on (an exception is propagating): every process of the group stops waiting for its frames"""
        self.failed = True
        self.N.lib().sfx_shm_abort(self.handle)

    def finish(self) -> None:
        """Every rank enters: flush (rank 0: until the writer has handed over the last frame), then ONE collective that carries each
        rank's outcome — so that a late failure on one rank (the sink's EPIPE on rank 0 after a peer has flushed everything) raises
        on every rank at once instead of leaving the healthy ones alone in a barrier — and only then the unmap."""
        import torch.distributed as dist
        error: Optional[BaseException] = None
        if not self.failed:
            try:
                self.N.check(self.N.lib().sfx_shm_flush(self.handle))
                if self.rank == 0:
                    self.N.check(self.N.lib().sfx_shm_drain_wait(self.handle))
            except Exception as caught:
                error = caught
        mine = None if not (self.failed or error) else f"rank {self.rank}: {error or 'its producer raised'}"
        if mine:
            self.N.lib().sfx_shm_abort(self.handle)                 # the peers' pushes and the writer fail at once instead of timing out
        outcomes = [mine]
        if self.world > 1:
            outcomes = [None]*self.world
            dist.all_gather_object(outcomes, mine)                  # also the barrier: nobody unmaps while the writer still reads
        self.N.lib().sfx_shm_destroy(self.handle)
        if error is not None:
            raise error
        failed = [outcome for outcome in outcomes if outcome]
        if failed and not self.failed:                              # (a rank whose own exception is propagating re-raises that one)
            raise RuntimeError("sharded export failed: " + "; ".join(failed))


def interleaved_runs(world: int, batches: list[tuple[int, int]]) -> list[tuple[int, int]]:
    """The sink's order for batches alternating between the ranks: (owner, count) per batch"""
    return [(index % world, count) for index, (_, count) in enumerate(batches)]


def interleaved_host_export(world: int, rank: int, batches: list[tuple[int, int]], advance, render, delivery, buffers: list) -> None:
    """Host mode: batch b is rendered by rank b % world into one of its `buffers` and pushed to the delivery queue; every rank
    ADVANCES through every batch (recurrences). A buffer is rendered into again only when the frames it held have left it.

        advance(first, count, buffer)   the tape / host state of those frames (all ranks, all batches; `buffer` is None for batches
                                        of other ranks — frame-loop scenes shade their own frames into it while they step)
        render(count, buffer)           render them into `buffer` (the owner), asynchronously on the context's stream
        delivery.push(buffer, count) / delivery.wait(frames)
    """
    marks: list[int] = []                                            # frames pushed after each of this rank's batches
    try:
        for index, (first, count) in enumerate(batches):
            if index % world != rank:
                advance(first, count, None)
                continue
            mine = len(marks)
            if mine >= len(buffers):
                delivery.wait(marks[mine - len(buffers)])           # the batch that last lived in this buffer has been copied out
            buffer = buffers[mine % len(buffers)]
            advance(first, count, buffer)
            render(count, buffer)
            delivery.push(buffer, count)
            marks.append(delivery.pushed)
    except BaseException:
        delivery.abort()                                            # a producer that raises tells the others now, not after their time-out
        raise


# ---- device mode: contiguous ranges, resident in HBM, sent to rank 0 ---------------------------------------------------------

class RangeTransfer:
    """Point-to-point transfer of frame chunks to rank 0 (RCCL send/recv; staged through host memory under gloo, which has no
    device transport — tests with several processes on one GPU). Rank 0 posts every receive up front, into the places of its
    resident buffer where the frames belong; a rank's chunks arrive in the order it sends them. (Up front on purpose: an RCCL send
    is a kernel that waits for its receive, so a receive posted late would stall the SENDER's stream and with it its renders. The
    price: a chunk that arrives later than the process group's time-out — NCCL's watchdog, 10 minutes by default — aborts the job;
    exports that long should raise `timeout=` in init_process_group or use SHADERFLOW_SHARD=device-sdma / host, which post nothing.)"""

    def __init__(self, world: int, rank: int, device):
        import torch
        import torch.distributed as dist
        self.world, self.rank = world, rank
        self.device = torch.device(device)
        self.staged = (dist.get_backend() == "gloo") and (self.device.type != "cpu")
        self.works: dict[tuple[int, int], tuple] = {}               # (source rank, chunk) → (work, staging tensor or None, target view)
        self.sent: list = []

    def expect(self, source: int, chunk: int, view) -> None:
        """Rank 0: chunk `chunk` of rank `source` lands in `view` (a slice of the resident buffer)"""
        import torch
        import torch.distributed as dist
        stage = torch.empty(view.numel(), dtype=view.dtype) if self.staged else None
        work = dist.irecv(stage if self.staged else view, src=source)
        self.works[(source, chunk)] = (work, stage, view)

    def send(self, view, first_frame: Optional[int] = None) -> None:
        import torch.distributed as dist
        tensor = view.cpu() if self.staged else view
        self.sent.append((dist.isend(tensor, dst=0), tensor))

    def arrived(self, source: int, chunk: int) -> None:
        """Rank 0: block until that chunk is in place"""
        work, stage, view = self.works.pop((source, chunk))
        work.wait()
        if stage is not None:
            view.copy_(stage)
        if self.device.type != "cpu":
            import torch
            torch.cuda.current_stream(self.device).synchronize()    # RCCL's wait only orders torch's stream; the read-out ring has its own

    def finish(self) -> None:
        for work, _ in self.sent:
            work.wait()
        self.sent.clear()

    def abort(self) -> None:
        """This rank cannot go on: receives posted for its chunks end with the process group (torchrun tears the job down)"""
        self.sent.clear()


class DeviceArray:
    """A raw device allocation as something torch can view without copying (`torch.as_tensor(DeviceArray(ptr, n), device=…)`): IPC
    handles exist for whole allocations only, and torch's caching allocator hands out pieces of larger ones"""

    def __init__(self, pointer: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (pointer, False), "version": 2}


class PeerWindowsUnavailable(RuntimeError):
    """The collective preflight of SdmaTransfer failed on some rank: raised on EVERY rank, with every rank's reason"""


class SdmaTransfer:
    """RangeTransfer's interface over peer windows (SHADERFLOW_SHARD=device-sdma): a chunk travels as ONE asynchronous device-to-device
    copy on a copy stream of the sending rank — SDMA engines, the rank's own xGMI link, no compute units, concurrent with the render of
    the next chunk — straight into rank 0's resident buffer, which every rank has mapped through an IPC handle. Completion travels
    on the host side: a sender waits for its PREVIOUS copy before it queues the next one (by then long done) and tells rank 0 with a
    one-integer message over gloo."""

    def __init__(self, world: int, rank: int, context, frame_bytes: int, resident_pointer: Optional[int], probe_at: int = 0, probe_bytes: int = 0):
        """Opens the windows and PROVES them, collectively: every sending rank moves `probe_bytes` through its window to `probe_at` (a place
        of rank 0's buffer its own first chunk overwrites later), the outcomes are all-gathered, and on ANY failure every rank closes its
        mapping and raises PeerWindowsUnavailable — so that the caller falls back for the whole group (tape.py: RCCL point-to-point), not one
        rank inside a half-done export. The copies have run on ONE GPU only: what does not work between two real ones is
        found here, before a frame is rendered. They travel on HIP's copy streams unless SHADERFLOW_PEER=engine names the SDMA engines."""
        import torch.distributed as dist
        self.world, self.rank, self.context, self.frame_bytes = world, rank, context, frame_bytes
        self.control = None if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")     # RCCL carries device tensors only
        self.window = None
        self.notices: list = []
        self.pending: Optional[int] = None
        self.chunks = 0
        problem = ""
        handle = [None]
        try:
            if rank == 0:
                handle = [context.peer_export(resident_pointer)]
        except Exception as error:                                   # noqa: BLE001 — whatever it is, the other ranks must hear of it
            problem = f"rank 0 could not export its buffer: {error}"
        dist.broadcast_object_list(handle, src=0, group=self.control)
        probe = None
        try:
            if rank != 0 and not problem:
                if handle[0] is None:
                    problem = "no window handle from rank 0"
                else:
                    self.window = context.peer_open(handle[0])
                    if os.environ.get("SHADERFLOW_PEER_INJECT") == f"preflight:{rank}":
                        raise RuntimeError("injected: this rank's peer copies do not work")     # tests: every rank must fall back together
                    if probe_bytes > 0:
                        probe = context.alloc(probe_bytes)
                        context.synchronize()
                        context.peer_copy(self.window + probe_at, probe, probe_bytes, lane=15)
                        context.peer_flush()
        except Exception as error:                                   # noqa: BLE001
            problem = f"rank {rank}: {error}"
        finally:
            if probe is not None:
                context.free(probe)
        problems = [None]*world
        dist.all_gather_object(problems, problem, group=self.control)
        failed = "; ".join(text for text in problems if text)
        if failed:
            try:
                if self.window is not None:
                    context.peer_close(self.window)
            except Exception:                                        # noqa: BLE001 — the first failure is the one to report
                pass
            self.window = None
            dist.barrier(group=self.control)                         # every peer has closed its mapping before rank 0 frees the buffer
            raise PeerWindowsUnavailable(failed)

    def expect(self, source: int, chunk: int, view) -> None:
        pass                                                        # nothing to post: the sender writes into place

    def _notify(self) -> None:
        import torch
        import torch.distributed as dist
        if self.pending is not None:
            self.context.peer_flush()                               # the previous chunk has landed in rank 0's HBM
            message = torch.tensor([self.pending], dtype=torch.int64)
            self.notices.append((dist.isend(message, dst=0, group=self.control), message))
            self.pending = None

    def send(self, view, first_frame: Optional[int] = None) -> None:
        self._notify()
        self.context.peer_copy(self.window + first_frame*self.frame_bytes, view.data_ptr(), view.numel(), lane=self.chunks % 16)
        self.pending = self.chunks
        self.chunks += 1

    ABORTED = -1                                                    # a sender that raised says so instead of the next chunk number

    def arrived(self, source: int, chunk: int) -> None:
        import torch
        import torch.distributed as dist
        message = torch.zeros(1, dtype=torch.int64)
        dist.recv(message, src=source, group=self.control)
        if int(message.item()) == self.ABORTED:
            raise RuntimeError(f"peer window: rank {source} failed before its chunk {chunk}")
        if int(message.item()) != chunk:
            raise RuntimeError(f"peer window: rank {source} reported chunk {int(message.item())}, expected {chunk}")

    def finish(self) -> None:
        self._notify()
        for work, _ in self.notices:
            work.wait()
        self.notices.clear()
        self.close()

    def abort(self) -> None:
        """This rank raised: rank 0, which waits for its next notice in arrived(), learns it now instead of never"""
        import torch
        import torch.distributed as dist
        self.pending = None
        if self.rank != 0:
            try:
                dist.send(torch.tensor([self.ABORTED], dtype=torch.int64), dst=0, group=self.control)
            except Exception:
                pass

    def close(self) -> None:
        if self.window is not None:
            self.context.peer_flush()
            self.context.peer_close(self.window)
            self.window = None


def contiguous_device_export(world: int, rank: int, total: int, batch: int, frame_bytes: int, advance, render, emit, resident, transfer: "RangeTransfer") -> None:
    """Device mode: rank r renders frames shard_frames(total, world, r) into `resident` and sends them to rank 0.

        advance(first, count)            tape / host state (a rank replays everything before its range without rendering)
        render(first, count, view)       render those frames into `view` (a slice of `resident`), complete when it returns
        emit(view, count)                rank 0: hand `count` frames to the sink
        resident                         rank 0: uint8 tensor of total*frame_bytes; other ranks: of their own range
    """
    ranges = [shard_frames(total, world, r) for r in range(world)]
    first, last = ranges[rank]
    if rank == 0:
        for source in range(1, world):
            for chunk, (f, c) in enumerate(shard_batches(*ranges[source], batch)):
                transfer.expect(source, chunk, resident[f*frame_bytes:(f + c)*frame_bytes])
    for f, c in shard_batches(0, first, batch):
        advance(f, c)                                               # the recurrences up to this rank's first frame
    base = 0 if rank == 0 else first
    for f, c in shard_batches(first, last, batch):
        advance(f, c)
        view = resident[(f - base)*frame_bytes:(f - base + c)*frame_bytes]
        render(f, c, view)
        if rank == 0:
            emit(view, c)
        else:
            transfer.send(view, f)
    if rank == 0:
        for source in range(1, world):
            for chunk, (f, c) in enumerate(shard_batches(*ranges[source], batch)):
                transfer.arrived(source, chunk)
                emit(resident[f*frame_bytes:(f + c)*frame_bytes], c)
    transfer.finish()
