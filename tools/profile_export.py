#!/usr/bin/env python3
"""Where the host-inclusive export of bench.py (`export_host`: scene.main() of 3 600 C3 frames to /dev/null) spends its time beyond
the PCIe read-out, and what differs between a plain process (≈ 2 080 frames/s) and bench.py's second leg (1 780-1 860): the same
export on a context of its own / on a torch stream / beside resident torch buffers. GPU box only."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from examples.scenes import Visualizer, make  # noqa: E402
from shaderflow_amd import _native as N  # noqa: E402
from shaderflow_amd import synth  # noqa: E402

pcm = synth.sweep_clip(60.0, 44100)
background = synth.background_image(1920, 1080, seed=0)


def export(label, context=None):
    scene = make(Visualizer, audio=(pcm, 44100), background=background, context=context)
    torch.cuda.synchronize()
    started = time.perf_counter()
    scene.main(width=3840, height=2160, ssaa=2, fps=60.0, time=60.0, output="/dev/null")
    torch.cuda.synchronize()
    took = time.perf_counter() - started
    print(f"{label}: 3600 frames in {took:.3f} s = {3600/took:.1f} frames/s; copy streams (looked at, in series) = {scene.context.copy_streams()}", flush=True)


import os
case = sys.argv[1] if len(sys.argv) > 1 else "own"
if case == "own":
    for k in range(6):
        export(f"own context #{k}")
else:
    stream = torch.cuda.Stream(device=0)
    torch.cuda.set_stream(stream)
    context = N.Context(0, stream.cuda_stream)
    export("context on a torch stream (first)", context)
    export("context on a torch stream", context)
    export("another context of its own, after those")
    export("context on a torch stream", context)
    export("another context of its own, after those")
    export("context on a torch stream", context)
