#!/usr/bin/env python3
"""
Export one of the example scenes (examples/scenes.py) from the command line:

    python examples/run.py Visualizer --audio song.wav --width 3840 --height 2160 --ssaa 2 --output out.mp4
    python examples/run.py Life --time 10 --output life.rgb
    python -m torch.distributed.run --nproc-per-node 8 examples/run.py MotionBlur --output blur.mp4     # one process per GPU

Without `--audio` the audio scenes get a synthetic sine sweep; an output ending in .rgb/.raw (or any path when no ffmpeg
binary exists) receives raw rgb24 frames, rows bottom-up; other paths are encoded by ffmpeg (shaderflow_amd/ffmpeg.py).
"""
import argparse
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main() -> None:
    import examples.scenes as scenes
    from shaderflow_amd import synth
    from shaderflow_amd.scene import ShaderScene
    names = sorted(n for n, c in vars(scenes).items() if isinstance(c, type) and issubclass(c, ShaderScene) and not n.startswith("_") and c is not ShaderScene)
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("scene", choices=names)
    p.add_argument("--audio", type=Path, help="WAV file (audio scenes)")
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--fps", type=float, default=60.0)
    p.add_argument("--ssaa", type=float, default=1.0)
    p.add_argument("--subsample", type=int, default=2)
    p.add_argument("--time", type=float, default=10.0, help="seconds to render")
    p.add_argument("--output", default="out.rgb")
    args = p.parse_args()

    world, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cls = getattr(scenes, args.scene)
    audio = None
    if issubclass(cls, scenes._AudioScene):
        audio = args.audio if args.audio else (synth.sweep_clip(args.time, 44100), 44100)
    scene = scenes.make(cls, audio=audio, device=local_rank)
    started = time.perf_counter()
    result = scene.main(width=args.width, height=args.height, fps=args.fps, ssaa=args.ssaa, subsample=args.subsample, time=args.time, output=args.output)
    took = time.perf_counter() - started
    if int(os.environ.get("RANK", "0")) == 0:
        frames = round(args.time*args.fps)
        print(f"{args.scene}: {frames} frames {args.width}x{args.height} ssaa {args.ssaa} → {result} in {took:.2f} s ({frames/took:.1f} frames/s) on {world} GPU(s)")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
