#!/usr/bin/env python3
"""Frame-loop scenes (python logic between frames: Dynamics, MotionBlur, Life; batch=False forces the loop for the others): frames/s of
a whole export to /dev/null at 1920x1080 and where the host spends its time (cProfile, by own time). GPU box only.
usage: profile_frame_loop.py [scene …] [--profile]"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import examples.scenes as scenes  # noqa: E402
from shaderflow_amd import synth  # noqa: E402

names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["Dynamics", "MotionBlur", "Life", "Multipass"]
for name in names:
    kind = getattr(scenes, name)
    background = synth.background_image(1920, 1080, seed=0)

    def build():
        try:
            return scenes.make(kind, background=background)
        except TypeError:
            return scenes.make(kind)
    for attempt in range(2):
        scene = build()
        profile = cProfile.Profile() if ("--profile" in sys.argv and attempt == 1) else None
        started = time.perf_counter()
        if profile:
            profile.enable()
        import os
        more = {"buffers": int(os.environ["RING_SLOTS"])} if "RING_SLOTS" in os.environ else {}
        scene.main(width=1920, height=1080, ssaa=1, fps=60.0, time=20.0, output="/dev/null", batch=False, **more)
        if profile:
            profile.disable()
        took = time.perf_counter() - started
    print(f"{name}: 1200 frames in {took:.3f} s = {1200/took:.0f} frames/s ({took/1200*1e6:.0f} us per frame)", flush=True)
    if profile:
        pstats.Stats(profile).sort_stats("tottime").print_stats(28)
