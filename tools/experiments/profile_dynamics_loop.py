#!/usr/bin/env python3
"""cProfile of the frame loop over the Dynamics scene (python logic every frame) at 1920x1080, yuv420p to /dev/null so that the link does
not hide the host: where the per-frame python goes. GPU box only."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import examples.scenes as scenes  # noqa: E402
from shaderflow_amd import synth  # noqa: E402

for attempt in range(2):
    scene = scenes.make(scenes.Dynamics, background=synth.background_image(1920, 1080, seed=0))
    profiler = cProfile.Profile()
    started = time.perf_counter()
    if attempt:
        profiler.enable()
    scene.main(width=1920, height=1080, ssaa=1, fps=60.0, time=20.0, output="/dev/null", pixel_format="yuv420p")
    if attempt:
        profiler.disable()
    took = time.perf_counter() - started
    print(f"attempt {attempt}: 1200 frames in {took:.3f} s = {1200/took:.0f} frames/s ({took/1200*1e6:.0f} us per frame)", flush=True)
pstats.Stats(profiler).sort_stats("tottime").print_stats(28)
