"""
The sharded export with real renders: two processes of one `gloo` process group share the test GPU (RCCL does not allow two
ranks on one device; FrameGather stages device buffers through host memory under gloo) and run the product's own
multi-rank code paths — the tape's round-robin batches (tape.py) and the frame loop with temporal warm-up (scene.py) —
end to end; rank 0's output must be the single-process export, byte for byte.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(name: str):
    import examples.scenes as scenes
    from shaderflow_amd import synth
    if name == "Visualizer":
        return scenes.make(scenes.Visualizer, audio=(synth.sweep_clip(3.0, 44100), 44100), background=synth.background_image(160, 90, seed=2))
    return scenes.make(getattr(scenes, name), background=synth.background_image(120, 68, seed=4))


KW = {"Visualizer": dict(width=96, height=54, fps=60.0, ssaa=2, time=130/60), "MotionBlur": dict(width=64, height=36, fps=30.0, ssaa=1, time=70/30)}


def _rank(rank: int, world: int, port: int, name: str, path: str, top_down=None, mode="host", pixel_format=None):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", RANK=str(rank), WORLD_SIZE=str(world), SHADERFLOW_SHARD=mode,
                      SHADERFLOW_SHM_SLOTS="5")                       # fewer ring slots than a batch: the back-pressure path runs too
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _build(name).main(output=path, top_down=top_down, pixel_format=pixel_format, **KW[name])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["host", "device", "device-sdma"])
@pytest.mark.parametrize("name,top_down", [("Visualizer", None), ("MotionBlur", None), ("Visualizer", True), ("MotionBlur", True)])
def test_two_ranks_on_one_gpu_reproduce_the_single_process_export(tmp_path, name, top_down, mode):
    """Both delivery modes of the sharded export (parallel.py): "host" = every rank reads its own batches out into the shared-memory
    ring and rank 0's writer thread interleaves them; "device" = contiguous HBM-resident ranges sent to rank 0 (tape scenes) /
    gathered rounds (frame-loop scenes); "device-sdma" = the same ranges copied into rank 0's buffer through an IPC window on the
    copy engines (two processes mapping one allocation of the one test GPU; frame-loop scenes keep the gathered rounds).
    `top_down=True`: the row order an ffmpeg sink asks for must reach EVERY rank, not only the one that owns the sink
    (a rank that missed it would deliver its batches upside down)"""
    whole = _build(name).main(output=bytes, top_down=top_down, **KW[name])
    if top_down:
        frames = np.frombuffer(whole, np.uint8).reshape(-1, KW[name]["height"], KW[name]["width"], 3)
        assert np.array_equal(frames[:, ::-1], np.frombuffer(_build(name).main(output=bytes, **KW[name]), np.uint8).reshape(frames.shape))
    path = str(tmp_path/"sharded.rgb")
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, name, path, top_down, mode)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    sharded = open(path, "rb").read()
    assert len(sharded) == len(whole)
    assert sharded == whole, f"{np.count_nonzero(np.frombuffer(sharded, np.uint8) != np.frombuffer(whole, np.uint8))} bytes differ"


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["host", "device", "device-sdma"])
@pytest.mark.parametrize("name", ["Visualizer", "MotionBlur"])
def test_two_ranks_deliver_yuv420p_frames_converted_where_they_were_rendered(tmp_path, name, mode):
    """pixel_format="yuv420p" in every shard mode: the rank that rendered a batch converts it and the PLANAR
    frames — half the bytes — are what crosses the rank's link (shared-memory ring, RCCL / staged point-to-point, peer window). The
    file rank 0 writes is the single-process yuv420p export, byte for byte; tape scene and frame-loop scene."""
    whole = _build(name).main(output=bytes, pixel_format="yuv420p", **KW[name])
    frames = round(KW[name]["time"]*KW[name]["fps"])
    assert len(whole) == frames*KW[name]["width"]*KW[name]["height"]*3//2
    path = str(tmp_path/"sharded.yuv")
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, name, path, None, mode, "yuv420p")) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    sharded = open(path, "rb").read()
    assert len(sharded) == len(whole)
    assert sharded == whole, f"{np.count_nonzero(np.frombuffer(sharded, np.uint8) != np.frombuffer(whole, np.uint8))} bytes differ"


def _rank_without_sink(rank: int, world: int, port: int, fail_on: int, mode: str = "host"):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", RANK=str(rank), WORLD_SIZE=str(world), SHADERFLOW_SHARD=mode,
                      SHADERFLOW_SHM_SLOTS="5", SHADERFLOW_SHM_TIMEOUT="20")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = _build("Visualizer")
    if rank == fail_on:
        from shaderflow_amd.tape import FrameTape
        render, calls = FrameTape.render, []

        def broken(self, count, device_out, first_slot=0):
            calls.append(count)
            raise RuntimeError("injected producer failure")      # rank 1 owns the middle batch of three: its only render
            return render(self, count, device_out, first_slot)
        FrameTape.render = broken
    output = None
    if fail_on == SINK_DIES and rank == 0:
        from shaderflow_amd.exporting import ExportingHelper

        def popen(self, open_sink=True):                             # the sink: a pipe nobody reads (an encoder that exited) → EPIPE in the writer
            self.scene.context.output_top_down(bool(self.top_down))
            reader, writer = os.pipe()
            os.close(reader)
            self.fileno = writer
        ExportingHelper.popen = popen
    if fail_on == SINK_DIES or mode != "host":
        output = f"/tmp/shaderflow-test-{os.getpid()}.rgb"
    try:
        scene.main(freewheel=(output is None), output=output, **KW["Visualizer"])
    finally:
        dist.destroy_process_group()


SINK_DIES = 100


@pytest.mark.timeout(200)
@pytest.mark.parametrize("fail_on", [-1, 1, SINK_DIES])
def test_two_ranks_without_a_sink_and_with_a_failing_producer(fail_on):
    """ADVICE round 2. (a) No sink (a freewheeling run under torchrun, fileno None): rank 0's writer consumes and DISCARDS the frames
    of both ranks in frame order — before, it exited at once and every producer sat in sfx_shm_push until the time-out.
    (b) A producer that raises mid-export tells the group through the segment's `failed` flag: both processes end within seconds
    with an error instead of waiting 900 s for frames that never come; the segment's name is gone from /dev/shm either way.
    (c) ADVICE round 3: the SINK fails on rank 0 (EPIPE in the writer thread): the outcome travels in HostDelivery.finish's collective,
    so the peer that had already flushed all its frames raises too instead of sitting alone in a barrier."""
    import time
    ctx = mp.get_context("spawn")
    port = _free_port()
    before = set(os.listdir("/dev/shm"))
    started = time.monotonic()
    procs = [ctx.Process(target=_rank_without_sink, args=(r, 2, port, fail_on)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=150)
    took = time.monotonic() - started
    codes = [p.exitcode for p in procs]
    if fail_on < 0:
        assert codes == [0, 0], codes
    else:
        assert all(code not in (0, None) for code in codes), codes          # the failure reaches BOTH ranks …
        assert took < 90, f"{took:.0f} s: a rank waited for its time-out"    # … at once
    assert not {name for name in set(os.listdir("/dev/shm")) - before if name.startswith("shaderflow-")}


@pytest.mark.timeout(200)
def test_peer_window_export_with_a_failing_sender():
    """device-sdma mode, rank 1 raises in its first render. Rank 0 — waiting for rank 1's first notice — is told
    (SdmaTransfer.abort), both ranks close the window, rank 0 frees its whole-export allocation, and both processes end with an error
    within seconds instead of hanging in a gloo recv / the barrier."""
    import time
    ctx = mp.get_context("spawn")
    port = _free_port()
    started = time.monotonic()
    procs = [ctx.Process(target=_rank_without_sink, args=(r, 2, port, 1, "device-sdma")) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=150)
    codes = [p.exitcode for p in procs]
    assert all(code not in (0, None) for code in codes), codes
    assert time.monotonic() - started < 90


@pytest.mark.timeout(400)
def test_bench_with_two_ranks_over_peer_windows():
    """bench.py --gpus 2 with SHADERFLOW_SHARD=device-sdma: the steps of rank 1 reach rank 0's buffers as peer copies"""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, SHADERFLOW_DIST_BACKEND="gloo", SHADERFLOW_SHARD="device-sdma")
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), str(root/"bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--frames-per-step", "4", "--width", "384", "--height", "216", "--no-export"]
    out = subprocess.run(command, capture_output=True, text=True, timeout=360, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    record = json.loads([line for line in out.stdout.splitlines() if line.startswith("{")][-1])
    assert record["n_gpus"] == 2 and record["rccl_ranks"] == 2 and "sdma" in record["gather"]["backend"]
    assert record["per_rank"][1]["sent_GB_per_s"] > 0 and "SDMA" in record["config"]["parallelism"]


@pytest.mark.timeout(400)
def test_bench_goes_on_without_a_transport_that_fails_its_preflight():
    """The SDMA peer copies have only ever run on one GPU: if they do not work between two real ones, EVERY rank drops the transport
    together before any timed leg (a collective preflight) and the line says so — here rank 1's preflight is made to fail"""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, SHADERFLOW_DIST_BACKEND="gloo", SHADERFLOW_BENCH_INJECT="sdma-preflight")
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), str(root/"bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--frames-per-step", "4", "--width", "384", "--height", "216", "--no-export"]
    out = subprocess.run(command, capture_output=True, text=True, timeout=360, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    record = json.loads([line for line in out.stdout.splitlines() if line.startswith("{")][-1])
    assert "injected" in record["gather"]["sdma_dropped"] and record["gather"]["chosen"] == "p2p"
    assert {(leg["transport"], leg["payload"]) for leg in record["gather"]["legs"]} == {("p2p", "rgb24"), ("p2p", "yuv420p")}
    assert record["value"] > 0 and record["rccl_ranks"] == 2


@pytest.mark.timeout(400)
def test_bench_with_two_ranks_prints_one_line_for_the_whole_job():
    """bench.py's N > 1 path as the driver launches it (torch.distributed.run, one rank per GPU), here with two ranks on the one
    test GPU over gloo: barrier + max-over-ranks timing, the gather to rank 0, ONE JSON line from rank 0 with the aggregate"""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, SHADERFLOW_DIST_BACKEND="gloo")
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), str(root/"bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--frames-per-step", "4", "--width", "384", "--height", "216"]
    out = subprocess.run(command, capture_output=True, text=True, timeout=360, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [line for line in out.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    record = json.loads(lines[0])
    assert record["n_gpus"] == 2 and record["steps"] == 2 and record["scaling"] == "weak"
    assert record["config"]["global_frames_per_step"] == 8 and record["value"] > 0 and record["config"]["ranks"] == 2
    assert "cpu_baseline" not in record                                       # rank 0 at N = 1 only
    # a multi-rank line is diagnosable: every rank took part, each one's render rate and what it sent to rank 0
    assert record["rccl_ranks"] == 2 and [r["rank"] for r in record["per_rank"]] == [0, 1]
    assert all(r["render_frames_per_s"] > 0 for r in record["per_rank"]) and record["per_rank"][1]["sent_GB_per_s"] > 0
    # no pin (the driver's launch): both transports x both payloads are measured as legs; the headline is the faster rgb24 leg
    legs = {(leg["transport"], leg["payload"]): leg for leg in record["gather"]["legs"]}
    assert set(legs) == {("p2p", "rgb24"), ("sdma", "rgb24"), ("p2p", "yuv420p"), ("sdma", "yuv420p")} and all(leg["value"] > 0 for leg in legs.values())
    assert record["gather"]["chosen"] in ("p2p", "sdma") and record["value"] == max(legs[("p2p", "rgb24")]["value"], legs[("sdma", "rgb24")]["value"])
    assert legs[("sdma", "rgb24")]["per_rank"][1]["peer_copies"]["copies"] > 0 and legs[("sdma", "rgb24")]["per_rank"][0]["peer_copies"]["copies"] == 0
    assert record["value_host"] == record["export_host"]["value"] and record["export_host"]["yuv420p"]["value"] > 0
    # the second number: the same frames through a real sharded export, read-out to host memory included (host mode)
    assert record["export_host"]["frames"] == 16 and record["export_host"]["value"] > 0 and "SHADERFLOW_SHARD=host" in record["export_host"]["mode"]
    assert record["roofline"]["bound"] == "valu" and record["roofline"]["kernel"].startswith("k_") and record["roofline"]["hbm"]["achieved"] > 0


@pytest.mark.timeout(500)
def test_plain_bench_command_with_gpus_2_becomes_two_ranks():
    """`python bench.py --gpus 2` with NO launcher around it — how the driver starts its 1-GPU bench, and how a first hardware scaling
    run might be started: the process must BECOME two ranks (a child torch.distributed.run, started before anything touches the GPU)
    and relay rank 0's line, never print a silent `n_gpus: 1` (missing 1)"""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(SHADERFLOW_DIST_BACKEND="gloo")
    command = [sys.executable, str(root/"bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames-per-step", "4", "--width", "384", "--height", "216", "--no-export"]
    out = subprocess.run(command, capture_output=True, text=True, timeout=420, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [line for line in out.stdout.splitlines() if line.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]           # the JSON line is the only — so the last — stdout line
    record = json.loads(lines[0])
    assert record["n_gpus"] == 2 and record["rccl_ranks"] == 2 and record["config"]["ranks"] == 2 and record["value"] > 0
    assert "without a launcher" in out.stderr


def _rank_with_env(rank: int, world: int, port: int, name: str, path: str, mode: str, extra: dict):
    os.environ.update(extra)
    _rank(rank, world, port, name, path, None, mode)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("extra", [{"SHADERFLOW_PEER_INJECT": "preflight:1"}, {"SHADERFLOW_PEER": "engine"}], ids=["preflight-fails", "named-engines"])
def test_peer_windows_are_proved_collectively_before_the_first_frame(tmp_path, extra, capfd):
    """SdmaTransfer moves a probe through every window and all-gathers the outcome BEFORE anything is rendered. With rank
    1's preflight made to fail every rank falls back to RCCL-style point-to-point together and the export is still the single-process
    one, byte for byte; with SHADERFLOW_PEER=engine (the SDMA engines named through HSA — opt-in since round 6, HIP's copy streams are
    the default) the windows are used as before."""
    whole = _build("Visualizer").main(output=bytes, **KW["Visualizer"])
    path = str(tmp_path/"sharded.rgb")
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rank_with_env, args=(r, 2, port, "Visualizer", path, "device-sdma", extra)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert open(path, "rb").read() == whole
    printed = capfd.readouterr().out
    assert ("falls back to device mode" in printed) == ("SHADERFLOW_PEER_INJECT" in extra), printed[-500:]
