#!/usr/bin/env python3
"""bench.py's two legs in one process (timed tape steps, then scene.main() to /dev/null on the SAME context on a torch stream), with
the wall-clock time of every batch hand-off of the export: is its first-export penalty a start-up cost or a rate? GPU box only."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch  # noqa: E402

from examples.scenes import Visualizer, make  # noqa: E402
from shaderflow_amd import _native as N  # noqa: E402
from shaderflow_amd import synth  # noqa: E402
from shaderflow_amd.exporting import ExportingHelper  # noqa: E402
from shaderflow_amd.message import ShaderMessage  # noqa: E402
from shaderflow_amd.tape import FrameTape  # noqa: E402

w, h, s, fpb = 3840, 2160, 2, 60
pcm = synth.sweep_clip(60.0, 44100)
background = synth.background_image(1920, 1080, seed=0)
stream = torch.cuda.Stream(device=0)
torch.cuda.set_stream(stream)
context = N.Context(0, stream.cuda_stream)
if "--early-streams" in sys.argv:
    print("copy streams chosen at context creation:", context.copy_streams())
if "--first-leg" in sys.argv:
    scene = make(Visualizer, audio=(pcm, 44100), background=background, context=context)
    scene.initialize()
    scene.exporting = scene.freewheel = scene.headless = True
    scene.realtime = False
    scene.fps, scene.subsample, scene.time = 60.0, 2, 0.0
    scene.relay(ShaderMessage.Shader.Compile)
    scene.resize(width=w, height=h)
    for module in scene.modules:
        module.setup()
    scene.set_duration(60.0)
    scene.ssaa = s
    tape = FrameTape(scene, batch=fpb).prepare(600)
    tape.bind_static_uniforms()
    buffers = [torch.zeros(fpb*w*h*3, dtype=torch.uint8, device="cuda") for _ in range(2)]
    steps = 2 if "--short" in sys.argv else 10
    for i in range(steps):
        tape.build(i*fpb, fpb)
        tape.render(fpb, buffers[i % 2].data_ptr())
    torch.cuda.synchronize()
    tape.release()
    if "--drop" in sys.argv:
        del buffers, tape, scene
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    if "--no-buffers" in sys.argv:
        del buffers
        torch.cuda.empty_cache()

marks = []
original = ExportingHelper.pipe_device_frames


def timed(self, *args, **kwargs):
    marks.append(time.perf_counter())
    return original(self, *args, **kwargs)


ExportingHelper.pipe_device_frames = timed
for attempt in range(2):
    marks.clear()
    scene2 = make(Visualizer, audio=(pcm, 44100), background=background, context=context)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scene2.main(width=w, height=h, ssaa=s, fps=60.0, time=60.0, output="/dev/null")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    gaps = [round((b - a)*1e3, 1) for a, b in zip(marks, marks[1:])]
    print(f"export {attempt}: {3600/(t1 - t0):.1f} frames/s; first hand-off after {(marks[0] - t0)*1e3:.0f} ms, last hand-off → end {(t1 - marks[-1])*1e3:.0f} ms; "
          f"ms between hand-offs: first five {gaps[:5]}, median {sorted(gaps)[len(gaps)//2]}, max {max(gaps)}; copy streams {context.copy_streams()}", flush=True)
