#!/usr/bin/env python3
"""
BASELINE config 5: offline export sharded over the GPUs of one node, frames gathered to rank 0 over RCCL/xGMI.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/export_sharded.py \
         --width 3840 --height 2160 --ssaa 2 --seconds 60 --output /dev/null

One process per GPU. Every rank builds the same scene on its own device and calls `scene.main(...)`; the frame tape
notices the process group, renders batch b on rank b % world, and rank 0 receives the finished frames in order and
writes them (raw rgb24, rows bottom-up, or through ffmpeg when a binary is present). Works with 1 process as well.
"""
import argparse
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--width", type=int, default=3840)
    p.add_argument("--height", type=int, default=2160)
    p.add_argument("--ssaa", type=float, default=2)
    p.add_argument("--seconds", type=float, default=10.0)
    p.add_argument("--output", default="/dev/null")
    args = p.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC (RCCL across processes on this pool)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from examples.scenes import Visualizer, make
    from shaderflow_amd import synth
    scene = make(Visualizer, audio=(synth.sweep_clip(args.seconds, 44100), 44100), background=synth.background_image(1920, 1080), device=local_rank)
    start = time.perf_counter()
    scene.main(width=args.width, height=args.height, ssaa=args.ssaa, fps=60, time=args.seconds, output=args.output)
    took = time.perf_counter() - start
    if int(os.environ.get("RANK", "0")) == 0:
        frames = round(args.seconds*60)
        print(f"{frames} frames {args.width}x{args.height} ssaa {args.ssaa} on {world} GPU(s): {took:.2f} s = {frames/took:.1f} frames/s "
              f"({args.seconds/took:.2f}x real time), host read-out included")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
