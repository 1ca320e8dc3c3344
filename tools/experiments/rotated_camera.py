#!/usr/bin/env python3
"""C3 (3840x2160, 2x SSAA) with a camera rolled about its forward axis (camera.py rotate2d): which kernel runs, how many of its blocks
miss their LDS tile, and how fast — the strip kernel admits axis-aligned cameras only. GPU box only."""
import math
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np  # noqa: E402

from shaderflow_amd import _native as N  # noqa: E402
from shaderflow_amd import synth  # noqa: E402
from tests.helpers import Gpu, gpu_bind_all, visualizer_inputs  # noqa: E402

w, h, ssaa = 3840, 2160, 2
gpu = Gpu()
u, arrays, params = visualizer_inputs(w, h, seed=1, volume=0.8, bg_size=(1920, 1080))
arrays["background"] = np.ascontiguousarray(np.flipud(synth.background_image(1920, 1080)))
u.iSSAA = float(ssaa)
import os
for degrees in [float(d) for d in os.environ.get('DEGREES', '0,5,17,45,90').split(',')]:
    c, s = math.cos(math.radians(degrees)), math.sin(math.radians(degrees))
    u.iCameraRight[0], u.iCameraRight[1], u.iCameraRight[2] = c, s, 0.0
    u.iCameraUpward[0], u.iCameraUpward[1], u.iCameraUpward[2] = -s, c, 0.0
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    target = gpu.empty(w, h, 3)
    gpu.ctx.tile_misses()
    N.check(gpu.lib.sfx_render_resolve(prog, target, ssaa, 2))
    misses = gpu.ctx.tile_misses()
    for attempt in range(3):
        gpu.ctx.synchronize()
        started = time.perf_counter()
        for _ in range(20):
            N.check(gpu.lib.sfx_render_resolve(prog, target, ssaa, 2))
        gpu.ctx.synchronize()
        took = time.perf_counter() - started
    print(f"camera rotated by {degrees:4.1f} degrees: {gpu.lib.sfx_last_kernel().decode()}  {20/took:7.1f} frames/s ({took/20*1e3:.2f} ms per frame), {misses} blocks off their tile", flush=True)
