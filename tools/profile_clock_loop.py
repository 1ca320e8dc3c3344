#!/usr/bin/env python3
"""Scenes in which only the clock moves between frames, exported at 1920x1080 to /dev/null: scene.next's loop (SHADERFLOW_CLOCK_LOOP=0)
against clockloop.ClockLoop (the default). GPU box only."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import examples.scenes as scenes  # noqa: E402
from shaderflow_amd import synth  # noqa: E402

for name in ("MotionBlur", "Life", "Multipass"):
    for clock_loop in (False, True):
        for attempt in range(2):
            scene = scenes.make(getattr(scenes, name), background=synth.background_image(1920, 1080, seed=0)) if name != "Life" else scenes.Life()
            scene.clock_loop = clock_loop
            started = time.perf_counter()
            scene.main(width=1920, height=1080, ssaa=1, fps=60.0, time=20.0, output="/dev/null")
            took = time.perf_counter() - started
        print(f"{name:10s} {'clock loop' if clock_loop else 'frame loop'}: 1200 frames in {took:.3f} s = {1200/took:.0f} frames/s ({took/1200*1e6:.0f} us per frame)", flush=True)
