#!/usr/bin/env python3
"""The frame loop's export ceiling: Life / Dynamics / MotionBlur at 1920x1080 to /dev/null as rgb24 (6.2 MB per frame over PCIe: 8 900 frames/s
at the 55.3 GB/s measured) and as yuv420p (3.1 MB: 17 800). GPU box only."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import examples.scenes as scenes  # noqa: E402
from shaderflow_amd import synth  # noqa: E402

for name in ("Life", "Dynamics", "MotionBlur"):
    for pixel_format in ("rgb24", "yuv420p"):
        for attempt in range(2):
            scene = scenes.Life() if name == "Life" else scenes.make(getattr(scenes, name), background=synth.background_image(1920, 1080, seed=0))
            started = time.perf_counter()
            scene.main(width=1920, height=1080, ssaa=1, fps=60.0, time=20.0, output="/dev/null", pixel_format=pixel_format)
            took = time.perf_counter() - started
        print(f"{name:10s} {pixel_format:8s}: 1200 frames in {took:.3f} s = {1200/took:.0f} frames/s ({took/1200*1e6:.0f} us per frame)", flush=True)
