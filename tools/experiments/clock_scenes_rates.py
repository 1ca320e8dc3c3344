#!/usr/bin/env python3
"""Steady-state export rates of the layered / temporal scenes at 1920x1080 to /dev/null (round 5: layered_fast.hpp + the native clock
sequence). A 20 s export carries ~15-25 ms of set-up (compile, ring allocation, first launches), which at 10 000+ frames/s is a fifth of
the run: the rate is the SLOPE between a 20 s and a 120 s export, beside the whole-call rates. SHADERFLOW_CLOCK_SEQUENCE=0: python between
the frames. GPU box only."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import examples.scenes as scenes  # noqa: E402
from shaderflow_amd import synth  # noqa: E402


def export(name: str, pixel_format: str, seconds: float) -> float:
    best = None
    for attempt in range(2):
        scene = scenes.Life() if name == "Life" else scenes.make(getattr(scenes, name), background=synth.background_image(1920, 1080, seed=0))
        started = time.perf_counter()
        scene.main(width=1920, height=1080, ssaa=1, fps=60.0, time=seconds, output="/dev/null", pixel_format=pixel_format)
        took = time.perf_counter() - started
        best = took if best is None else min(best, took)
    return best


names = sys.argv[1:] or ["Multipass", "MotionBlur", "Life"]
print(f"# SHADERFLOW_CLOCK_SEQUENCE={os.environ.get('SHADERFLOW_CLOCK_SEQUENCE', '1')}  SHADERFLOW_LAYERED_FAST={os.environ.get('SHADERFLOW_LAYERED_FAST', '1')}")
for name in names:
    for pixel_format in ("rgb24", "yuv420p"):
        short, long = export(name, pixel_format, 20.0), export(name, pixel_format, 120.0)
        slope = (long - short)/6000.0
        print(f"{name:10s} {pixel_format:8s}: 1200 frames {short*1e3:7.1f} ms ({1200/short:7.0f} frames/s), 7200 frames {long*1e3:7.1f} ms ({7200/long:7.0f} frames/s); "
              f"steady state {slope*1e6:6.1f} us per frame = {1/slope:7.0f} frames/s", flush=True)
