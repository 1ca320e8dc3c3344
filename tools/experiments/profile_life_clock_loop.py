#!/usr/bin/env python3
"""cProfile of clockloop.ClockLoop over the Life scene at 1920x1080 (1 200 frames to /dev/null): which native calls the 142 us per frame are.
GPU box only."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import examples.scenes as scenes  # noqa: E402

for attempt in range(2):
    scene = scenes.Life()
    profiler = cProfile.Profile()
    started = time.perf_counter()
    if attempt:
        profiler.enable()
    scene.main(width=1920, height=1080, ssaa=1, fps=60.0, time=20.0, output="/dev/null")
    if attempt:
        profiler.disable()
    took = time.perf_counter() - started
    print(f"attempt {attempt}: 1200 frames in {took:.3f} s = {1200/took:.0f} frames/s ({took/1200*1e6:.0f} us per frame)", flush=True)
pstats.Stats(profiler).sort_stats("tottime").print_stats(18)
