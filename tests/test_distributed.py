"""
The N > 1 path on CPU: frame-range sharding and the two-slot asynchronous gather of finished frames to rank 0,
with world_size 2 over gloo (the same code runs over RCCL on the GPUs). CPU only.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from shaderflow_amd.parallel import FrameGather, shard_batches, shard_frames


def test_shard_frames_covers_everything_once():
    for total in (1, 7, 60, 3600, 3601):
        for world in (1, 2, 3, 8):
            ranges = [shard_frames(total, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert shard_frames(3600, 8, 3) == (1350, 1800)               # BASELINE config 5
    assert shard_batches(1350, 1800, 60)[0] == (1350, 60) and sum(c for _, c in shard_batches(1350, 1800, 64)) == 450
    assert shard_batches(0, 0, 60) == []


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, nbytes: int, steps: int, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gather = FrameGather(world, rank, nbytes, torch.device("cpu"))
        buffers = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(2)]
        seen = []
        for step in range(steps):
            slot = step % 2
            gather.wait(slot)                                     # buffer reuse fence, as in bench.py
            if rank == 0 and step >= 2:
                seen.append([int(t[0]) for t in gather.frames(slot)])
            buffers[slot].fill_((17*step + rank) % 256)           # "render" this rank's batch
            gather.start(slot, buffers[slot])
        gather.wait_all()
        if rank == 0:
            for step in range(max(0, steps - 2), steps):
                seen.append([int(t[0]) for t in gather.frames(step % 2)])
            out.put(seen)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gather_over_gloo():
    world, steps, nbytes = 2, 5, 4096
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nbytes, steps, out)) for r in range(world)]
    for p in procs:
        p.start()
    seen = out.get(timeout=90)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    want = [[(17*step + rank) % 256 for rank in range(world)] for step in range(steps)]
    assert sorted(seen) == sorted(want), seen


def _export_worker(rank: int, world: int, port: int, total: int, batch: int, out):
    import numpy as np
    from shaderflow_amd.parallel import round_robin_export
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frame_bytes = 8
        buffers = [torch.zeros(batch*frame_bytes, dtype=torch.uint8) for _ in range(2)]
        gather = FrameGather(world, rank, batch*frame_bytes, torch.device("cpu"))
        state = {"tape": None, "seen": 0}                        # the "recurrence": every rank must advance through every frame in order
        emitted = []

        def advance(first, count):
            assert first == state["seen"], (rank, first, state["seen"])
            state["seen"] += count
            state["tape"] = (first, count)

        def render(count, buffer):
            first, n = state["tape"]
            frames = torch.arange(first, first + n, dtype=torch.int64).view(torch.uint8)     # frame k is the 8 bytes of int64(k)
            buffer[:n*frame_bytes] = frames

        def emit(buffer, count):
            emitted.extend(np.frombuffer(buffer[:count*frame_bytes].numpy().tobytes(), np.int64).tolist())

        round_robin_export(world, rank, shard_batches(0, total, batch), advance, render, emit, gather, buffers, frame_bytes)
        assert state["seen"] == total
        if rank == 0:
            out.put(emitted)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("total,batch", [(50, 8), (64, 8), (7, 4)])
def test_round_robin_export_emits_every_frame_in_order(total, batch):
    """The sharded export orchestration (batch b rendered by rank b % world, gathered per round) with fake renders"""
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_export_worker, args=(r, world, port, total, batch, out)) for r in range(world)]
    for p in procs:
        p.start()
    emitted = out.get(timeout=90)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert emitted == list(range(total))


def test_round_robin_export_single_rank():
    from shaderflow_amd.parallel import round_robin_export
    log = []
    buffers = [bytearray(4), bytearray(4)]
    round_robin_export(1, 0, shard_batches(0, 10, 4), lambda f, c: log.append(("advance", f, c)),
                       lambda c, b: log.append(("render", c)), lambda b, c: log.append(("emit", c)), None, buffers, 1)
    assert [e for e in log if e[0] == "emit"] == [("emit", 4), ("emit", 4), ("emit", 2)]
    assert [e for e in log if e[0] == "advance"] == [("advance", 0, 4), ("advance", 4, 4), ("advance", 8, 2)]


def test_frame_modes_cover_owned_batches_and_their_warmup():
    from shaderflow_amd.parallel import frame_modes
    batches = shard_batches(0, 50, 8)
    for world in (1, 2, 3):
        owners = np.zeros(50, int)
        for rank in range(world):
            modes = np.array(frame_modes(batches, world, rank, 3))
            owners += (modes == 2)
            for index, (first, count) in enumerate(batches):
                if index % world == rank:
                    assert (modes[first:first + count] == 2).all()
                    assert (modes[max(0, first - 3):first] >= 1).all()                 # warm-up rendered (or owned)
            assert ((modes == 1).sum() <= 3*len(batches))
        assert (owners == 1).all()                                                     # every frame kept by exactly one rank
    unbounded = np.array(frame_modes(batches, 2, 1, None))
    assert (unbounded[:40] >= 1).all() and (unbounded[48:] == 0).all()                 # everything before the rank's last batch
    assert frame_modes(batches, 1, 0, 5) == [2]*50


def _loop_worker(rank: int, world: int, port: int, total: int, batch: int, warmup: int, out):
    from shaderflow_amd.parallel import frame_modes, sharded_frame_loop
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frame_bytes = 8
        batches = shard_batches(0, total, batch)
        modes = frame_modes(batches, world, rank, warmup)
        buffers = [torch.zeros(batch*frame_bytes, dtype=torch.uint8) for _ in range(2)]
        gather = FrameGather(world, rank, batch*frame_bytes, torch.device("cpu"))
        history, walked, emitted = [], [], []

        def step(frame, mode, buffer, offset):
            walked.append(frame)                                                        # host logic runs for EVERY frame, in order
            if mode:
                history.append(frame)
            if mode == 2:                                                               # a kept frame needs its `warmup` predecessors rendered
                assert all(k in history for k in range(max(0, frame - warmup), frame)), (rank, frame)
            if mode == 2:
                buffer[offset:offset + frame_bytes] = torch.tensor([frame], dtype=torch.int64).view(torch.uint8)

        def emit(buffer, count):
            emitted.extend(np.frombuffer(buffer[:count*frame_bytes].numpy().tobytes(), np.int64).tolist())

        sharded_frame_loop(world, rank, batches, modes, step, lambda: None, emit, gather, buffers, frame_bytes)
        assert walked == list(range(total))
        if rank == 0:
            out.put(emitted)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("total,batch,warmup", [(50, 8, 3), (23, 4, 6)])
def test_sharded_frame_loop_over_gloo(total, batch, warmup):
    """Frame-loop sharding with temporal warm-up: every rank walks all frames, renders its batches + warm-up, rank 0 emits in order"""
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, world, port, total, batch, warmup, out)) for r in range(world)]
    for p in procs:
        p.start()
    emitted = out.get(timeout=90)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert emitted == list(range(total))


# ---- the two delivery modes of the sharded export (parallel.py) -------------------------------------------------------------------

def _device_mode_worker(rank: int, world: int, port: int, total: int, batch: int, out):
    """contiguous_device_export over gloo with stand-in renders: rank r renders shard_frames(total, world, r) into a resident buffer,
    sends it to rank 0 chunk by chunk (RangeTransfer: isend/irecv), rank 0 emits every frame in order"""
    from shaderflow_amd.parallel import RangeTransfer, contiguous_device_export
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frame_bytes = 8
        first, last = shard_frames(total, world, rank)
        resident = torch.zeros((total if rank == 0 else max(1, last - first))*frame_bytes, dtype=torch.uint8)
        state = {"seen": 0}
        rendered, emitted = [], []

        def advance(f, c):
            assert f == state["seen"], (rank, f, state["seen"])                        # the recurrences see every frame up to the range's end, in order
            state["seen"] += c

        def render(f, c, view):
            rendered.extend(range(f, f + c))
            view[:] = torch.arange(f, f + c, dtype=torch.int64).view(torch.uint8)

        def emit(view, c):
            emitted.extend(np.frombuffer(view.numpy().tobytes(), np.int64).tolist())

        contiguous_device_export(world, rank, total, batch, frame_bytes, advance, render, emit, resident, RangeTransfer(world, rank, torch.device("cpu")))
        assert rendered == list(range(first, last)) and state["seen"] == last          # ONE contiguous range per rank, nothing beyond it
        if rank == 0:
            out.put(emitted)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("total,batch", [(50, 8), (64, 8), (7, 4), (3, 4)])
def test_contiguous_device_export_over_gloo(total, batch):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_device_mode_worker, args=(r, world, port, total, batch, out)) for r in range(world)]
    for p in procs:
        p.start()
    emitted = out.get(timeout=90)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert emitted == list(range(total))


class _RecordingDelivery:
    """Stand-in for HostDelivery: records what a rank pushes and checks the buffer-reuse fence"""

    def __init__(self):
        self.pushed, self.frames, self.waited, self.live = 0, [], [], {}

    def push(self, buffer, count):
        assert id(buffer) not in self.live, "a buffer was rendered into again before its frames were waited for"
        self.live[id(buffer)] = self.pushed + count
        self.frames.extend(buffer[:count])
        self.pushed += count

    def wait(self, frames):
        self.waited.append(frames)
        self.live = {key: mark for key, mark in self.live.items() if mark > frames}


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("total,batch", [(50, 8), (16, 8), (5, 8)])
def test_interleaved_host_export_orders_frames_and_fences_buffers(world, total, batch):
    """Host mode: batch b on rank b % world, two buffers per rank; the writer's run list reassembles 0..total-1"""
    from shaderflow_amd.parallel import interleaved_host_export, interleaved_runs
    batches = shard_batches(0, total, batch)
    per_rank = []
    for rank in range(world):
        delivery, advanced, current = _RecordingDelivery(), [], {}
        buffers = [[None]*batch, [None]*batch]

        def advance(first, count, buffer, advanced=advanced, current=current):
            advanced.append((first, count, buffer is not None))
            current["first"] = first

        def render(count, buffer, current=current):
            buffer[:count] = range(current["first"], current["first"] + count)

        interleaved_host_export(world, rank, batches, advance, render, delivery, buffers)
        assert [(f, c) for f, c, _ in advanced] == batches                              # every rank advances through every batch, in order
        assert [own for _, _, own in advanced] == [index % world == rank for index in range(len(batches))]
        per_rank.append(delivery.frames)
    taken, order = [0]*world, []
    for owner, count in interleaved_runs(world, batches):                               # what rank 0's writer does with the rings
        order.extend(per_rank[owner][taken[owner]:taken[owner] + count])
        taken[owner] += count
    assert order == list(range(total))


def test_shm_ring_is_cut_to_the_free_space_of_its_filesystem(tmp_path, monkeypatch):
    """The cross-process frame ring lives in tmpfs; mapping more than the filesystem can back ends in SIGBUS, so the slot count is
    cut to half of the free space, and refused with a message when two frames per rank do not fit"""
    import os
    from shaderflow_amd import parallel

    class Stat:
        f_frsize = 4096

        def __init__(self, free_bytes):
            self.f_bavail = free_bytes//4096

    frame = 3840*2160*3
    monkeypatch.setattr(os, "statvfs", lambda path: Stat(64 << 30))
    assert parallel.shm_slots_that_fit(frame, 120, 8) == 120                       # 23.9 GB of ring inside 32 GB
    monkeypatch.setattr(os, "statvfs", lambda path: Stat(8 << 30))
    assert parallel.shm_slots_that_fit(frame, 120, 8) == (4 << 30)//(frame*8)      # cut to half of the free space
    monkeypatch.setattr(os, "statvfs", lambda path: Stat(64 << 20))                # a container's default /dev/shm
    with pytest.raises(RuntimeError, match="frame ring needs"):
        parallel.shm_slots_that_fit(frame, 120, 8)


# ---- bench.py as the driver may start it: `python bench.py --gpus N` with no launcher (missing 1) -------------------

def _bench_module():
    import importlib.util
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("bench_under_test", root/"bench.py")
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)                                 # (imports nothing heavy at module level: no torch, no HIP library)
    return module, root


def test_bench_launcher_command_is_the_drivers_own():
    bench, root = _bench_module()
    command = bench.launcher_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29513)
    assert command[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in command
    assert command[command.index("--master-addr") + 1] == "127.0.0.1" and command[command.index("--master-port") + 1] == "29513"
    at = command.index(str(root/"bench.py"))
    assert command[at + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]            # the ranks get the SAME arguments


def _run_bench(arguments, **env_changes):
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "SHADERFLOW_DIST_BACKEND")}
    env.update(env_changes)
    return subprocess.run([sys.executable, str(root/"bench.py"), *arguments], capture_output=True, text=True, timeout=280, cwd=root, env=env)


@pytest.mark.timeout(300)
def test_bench_refuses_one_rank_under_gpus_2():
    """WORLD_SIZE=1 with --gpus 2 (a launcher that started ONE rank): until round 5 this rendered on one GPU and printed `n_gpus: 1`"""
    out = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert out.returncode != 0 and out.stdout.strip() == "" and "--gpus 2 but WORLD_SIZE=1" in out.stderr
    out = _run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert out.returncode != 0 and out.stdout.strip() == "" and "--gpus 1 but WORLD_SIZE=2" in out.stderr


@pytest.mark.timeout(300)
def test_plain_bench_command_spawns_its_ranks_and_fails_loudly_without_gpus():
    """No launcher, --gpus 2, a node without (enough) GPUs: the parent starts the ranks as a child process, every rank refuses with one
    sentence, the parent exits non-zero and prints NO line. (On a GPU box the same command is tests/test_gpu_distributed.py's.)"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are visible: the command would run")
    out = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--frames-per-step", "2", "--width", "64", "--height", "36"])
    assert out.returncode != 0 and out.stdout.strip() == "", out.stdout[-500:]
    assert "without a launcher" in out.stderr and "needs 2 visible GPUs" in out.stderr and "no line" in out.stderr


# ---- the sharded export's orchestration at the north star's size: world 8, 3 601 frames (uneven ranges), a failing middle rank ------

def _spawn_world(target, world, args, timeout=150):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, *args, out)) for r in range(world)]
    for p in procs:
        p.start()
    return procs, out


@pytest.mark.timeout(240)
def test_contiguous_device_export_world_8_uneven_ranges():
    """BASELINE config 5's layout (3 600 frames over 8 ranks) plus one frame, so that the ranges are uneven: 451 + 7 x 450"""
    total, batch, world = 3601, 60, 8
    assert [shard_frames(total, world, r)[1] - shard_frames(total, world, r)[0] for r in range(world)] == [451] + [450]*7
    procs, out = _spawn_world(_device_mode_worker, world, (total, batch))
    emitted = out.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert emitted == list(range(total))


@pytest.mark.timeout(240)
def test_round_robin_export_world_8_uneven_rounds():
    total, batch, world = 3601, 60, 8                              # 61 batches: 7 full rounds of 8, a round of 5, the last batch of 1 frame
    procs, out = _spawn_world(_export_worker, world, (total, batch))
    emitted = out.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert emitted == list(range(total))


def _failing_device_worker(rank: int, world: int, port: int, total: int, batch: int, failing: int, out):
    """_device_mode_worker with rank `failing` raising in its third render: nobody may hang — the failing rank ends with its error, rank 0
    (which waits for that rank's chunks) with an error of its own, the others finish their sends"""
    from shaderflow_amd.parallel import RangeTransfer, contiguous_device_export
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    frame_bytes = 8
    first, last = shard_frames(total, world, rank)
    resident = torch.zeros((total if rank == 0 else max(1, last - first))*frame_bytes, dtype=torch.uint8)
    renders, emitted = [0], []

    def render(f, c, view):
        renders[0] += 1
        if rank == failing and renders[0] == 3:
            raise RuntimeError("injected: this rank's render failed")
        view[:] = torch.arange(f, f + c, dtype=torch.int64).view(torch.uint8)

    def emit(view, c):
        emitted.extend(np.frombuffer(view.numpy().tobytes(), np.int64).tolist())

    transfer = RangeTransfer(world, rank, torch.device("cpu"))
    outcome = "done"
    try:
        contiguous_device_export(world, rank, total, batch, frame_bytes, lambda f, c: None, render, emit, resident, transfer)
    except BaseException as error:                                  # noqa: BLE001 — what the export does with it: tape.py:361-363
        outcome = f"{type(error).__name__}: {error}"
        transfer.abort()
    out.put((rank, outcome, len(emitted), emitted == list(range(len(emitted)))))
    out.close(); out.join_thread()                                   # (the queue's feeder thread has written before the process ends)
    os._exit(0 if outcome == "done" else 1)                         # (no destroy_process_group: a peer is gone, as after a real failure)


@pytest.mark.timeout(300)
def test_contiguous_device_export_world_8_with_a_failing_middle_rank():
    total, batch, world, failing = 3601, 60, 8, 3
    procs, out = _spawn_world(_failing_device_worker, world, (total, batch, failing))
    reports = {}
    for _ in range(world):
        rank, outcome, count, in_order = out.get(timeout=200)
        reports[rank] = (outcome, count, in_order)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode is not None, "a rank hung"
    assert "injected" in reports[failing][0]
    assert reports[0][0] != "done", reports[0]                      # rank 0 cannot have emitted rank 3's frames: it must say so, not finish
    first_of_failing = shard_frames(total, world, failing)[0]
    # what rank 0 DID hand to the sink is a correct prefix: its own range, ranks 1-2, and the chunks rank 3 sent before it failed
    assert reports[0][2] and first_of_failing <= reports[0][1] <= first_of_failing + 2*batch, reports[0]
    for rank in (1, 2, 4, 5, 6, 7):
        assert reports[rank][0] == "done", (rank, reports[rank])
