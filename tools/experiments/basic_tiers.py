#!/usr/bin/env python3
"""default.glsl (the Basic scene) at 3840x2160 2xSSAA under cameras that put different tiers of k_separable_fused<default> on the
screen: zoom 1 (the bench's frame: disc, ring band, checkerboard), zoom 0.2 (everything deep inside the disc: the far field alone),
zoom 2.2 (checkerboard and a sliver of ring), zoom 0.74 (the ring band fills most of the frame). Time per frame of back-to-back
launches; run it under `rocprofv3 --kernel-trace --stats` for the kernel's own duration. GPU box only."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))

from shaderflow_amd import _native as N  # noqa: E402
from tests.helpers import Gpu, gpu_bind_all, visualizer_inputs  # noqa: E402

w, h, ssaa = 3840, 2160, 2
gpu = Gpu()
for zoom in [float(z) for z in os.environ.get("ZOOMS", "1,0.2,2.2,0.74").split(",")]:
    u, arrays, params = visualizer_inputs(w, h, seed=5)
    u.iSSAA, u.iTau, u.iCameraZoom = float(ssaa), 0.37, zoom
    prog, _ = gpu.program("default")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    target = gpu.empty(w, h, 3)
    count = int(os.environ.get("LAUNCHES", "400"))
    for attempt in range(3):
        gpu.ctx.synchronize()
        started = time.perf_counter()
        for _ in range(count):
            N.check(gpu.lib.sfx_render_resolve(prog, target, ssaa, 2))
        gpu.ctx.synchronize()
        took = time.perf_counter() - started
    print(f"zoom {zoom:5.2f}: {gpu.lib.sfx_last_kernel().decode()}  {took/count*1e6:7.1f} us per frame", flush=True)
