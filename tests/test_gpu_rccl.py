"""
The RCCL backend on the ONE test GPU: every other multi-rank test forces gloo, because RCCL refuses two
ranks on one device — so `init_process_group("nccl", device_id=…)`, collectives on device tensors, grouped point-to-point calls on the
render stream and the shard modes under an RCCL group had never executed before the driver's first 8-GPU run. Here they run in a
group of ONE rank (`SHADERFLOW_FORCE_DIST=1` makes bench.py and the exports take their multi-rank paths at world size 1), always in
child processes with their own time-outs.
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rccl_env(**extra) -> dict:
    env = dict(os.environ, SHADERFLOW_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    env.pop("SHADERFLOW_DIST_BACKEND", None)                         # the default backend: "nccl" = RCCL
    return env


@pytest.mark.timeout(500)
@pytest.mark.parametrize("shard", ["device", "device-sdma", None])
def test_bench_over_rccl_on_one_gpu(shard):
    """bench.py exactly as a rank of the driver's 8-GPU run executes it — RCCL group initialised with device_id, the render stream as
    torch's current stream, every piece of every step through `batch_isend_irecv` (to this rank itself: RCCL's point-to-point kernel
    between two device buffers), the all-reduces behind `rccl_ranks` and the max-over-ranks time, all_gather_object, barriers,
    destroy — and ONE JSON line. "device-sdma": the peer-window path (IPC export of rank 0's buffers) under the same group."""
    command = [sys.executable, str(ROOT/"bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--frames-per-step", "8",
               "--width", "640", "--height", "360", "--no-cpu-baseline"]
    out = subprocess.run(command, capture_output=True, text=True, timeout=420, cwd=ROOT, env=_rccl_env(**({"SHADERFLOW_SHARD": shard} if shard else {})))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [line for line in out.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1 and out.stdout.rstrip().endswith(lines[0]), out.stdout[-2000:]      # the line is the LAST thing on stdout
    record = json.loads(lines[0])
    assert record["n_gpus"] == 1 and record["rccl_ranks"] == 1 and record["value"] > 0
    assert record["per_rank"][0]["render_frames_per_s"] > 0
    legs = {(leg["transport"], leg["payload"]): leg for leg in record["gather"]["legs"]}
    if shard == "device":
        assert record["gather"]["backend"] == "nccl" and record["gather"]["loopback"] is True
        assert record["gather"]["loopback_intact"] is True             # RCCL delivered the rendered bytes
        assert record["per_rank"][0]["sent_GB_per_s"] > 0 and "over nccl" in record["config"]["parallelism"]
        assert set(legs) == {("p2p", "rgb24"), ("p2p", "yuv420p")} and legs[("p2p", "yuv420p")]["loopback_intact"] is True
    elif shard == "device-sdma":
        assert "sdma" in record["gather"]["backend"] and set(legs) == {("sdma", "rgb24"), ("sdma", "yuv420p")}
        # one rank: the copies run from this rank's frame buffers into its own receive buffers — the copier thread, the named engines
        # (or HIP's streams where HSA declines a same-device pair) and the fences, with the delivered bytes compared
        route = legs[("sdma", "rgb24")]["per_rank"][0]["peer_copies"]
        assert route["route"] in ("sdma-engines", "hip-streams") and route["copies"] > 0
        assert legs[("sdma", "rgb24")]["loopback_intact"] is True and legs[("sdma", "yuv420p")]["loopback_intact"] is True
    else:
        # the driver's launch: no pin — both transports and both payloads are measured as legs, the headline is the faster rgb24 leg
        assert set(legs) == {("p2p", "rgb24"), ("sdma", "rgb24"), ("p2p", "yuv420p"), ("sdma", "yuv420p")}
        assert record["gather"]["chosen"] in ("p2p", "sdma") and record["value"] == max(legs[("p2p", "rgb24")]["value"], legs[("sdma", "rgb24")]["value"])
        assert all(leg["value"] > 0 for leg in legs.values()) and record["value_yuv420p"] > 0
    assert record["export_host"]["yuv420p"]["value"] > 0                # the sharded export delivers planar frames too (converted on the rendering rank)
    # the export leg ran the sharded export's host mode (shared-memory ring, writer thread) as the only rank of the RCCL group
    assert record["export_host"]["value"] > 0 and record["export_host"]["frames"] == 24


def _export_rank(port: int, name: str, path: str, mode: str, pixel_format=None):
    import torch
    import torch.distributed as dist

    from tests.test_gpu_distributed import KW, _build
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", RANK="0", WORLD_SIZE="1", SHADERFLOW_SHARD=mode,
                      SHADERFLOW_FORCE_DIST="1", SHADERFLOW_SHM_SLOTS="5", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        _build(name).main(output=path, pixel_format=pixel_format, **KW[name])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("pixel_format", [None, "yuv420p"])
@pytest.mark.parametrize("mode", ["host", "device", "device-sdma"])
@pytest.mark.parametrize("name", ["Visualizer", "MotionBlur"])
def test_sharded_export_paths_under_an_rccl_group_of_one(tmp_path, name, mode, pixel_format):
    """The sharded export's code paths with RCCL as the group's backend: tape scenes through `contiguous_device_export` (the resident
    torch buffer, rank 0's verdict broadcast as a device tensor, RangeTransfer / the peer window + its gloo side group under RCCL) or
    the shared-memory ring; frame-loop scenes through `FrameGather` (`dist.gather` of device buffers — a real RCCL collective — with the
    context's stream ordered against torch's). The file must be the single-process export, byte for byte — as rgb24, and as yuv420p
    converted on the rendering rank."""
    from tests.test_gpu_distributed import KW, _build
    whole = _build(name).main(output=bytes, pixel_format=pixel_format, **KW[name])
    path = str(tmp_path/"sharded.rgb")
    process = mp.get_context("spawn").Process(target=_export_rank, args=(_free_port(), name, path, mode, pixel_format))
    process.start()
    process.join(timeout=240)
    if process.is_alive():
        process.kill()
        pytest.fail("the export under RCCL did not finish")
    assert process.exitcode == 0, f"exit code {process.exitcode}"
    sharded = open(path, "rb").read()
    assert len(sharded) == len(whole)
    assert sharded == whole, f"{np.count_nonzero(np.frombuffer(sharded, np.uint8) != np.frombuffer(whole, np.uint8))} bytes differ"
