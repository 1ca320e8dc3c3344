#!/usr/bin/env python3
"""Experiment (round 4): the fused kernel storing its RGB8 frames STRAIGHT into pinned host memory (no device frame buffer, no copy
engine, no copy stream) against render-to-HBM + SDMA read-out. GPU box only.  usage: render_to_host.py [frames_per_launch]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch  # noqa: E402

from examples.scenes import Visualizer, make  # noqa: E402
from shaderflow_amd import synth  # noqa: E402
from shaderflow_amd.message import ShaderMessage  # noqa: E402
from shaderflow_amd.tape import FrameTape  # noqa: E402

fpb = int(sys.argv[1]) if len(sys.argv) > 1 else 60
w, h, s = 3840, 2160, 2
pcm = synth.sweep_clip(60.0, 44100)
scene = make(Visualizer, audio=(pcm, 44100), background=synth.background_image(1920, 1080, seed=0))
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
tape = FrameTape(scene, batch=fpb).prepare(3600)
tape.bind_static_uniforms()
frame_bytes = w*h*3
context = scene.context
device = [torch.zeros(fpb*frame_bytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
host = [torch.zeros(fpb*frame_bytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
torch.cuda.synchronize()
steps = 3600//fpb


def run(targets, label):
    for warm in range(2):
        tape.build(0, fpb)
        tape.render(fpb, targets[warm % 2].data_ptr())
    context.synchronize()
    started = time.perf_counter()
    for i in range(steps):
        tape.build(i*fpb, fpb)
        tape.render(fpb, targets[i % 2].data_ptr())
    context.synchronize()
    took = time.perf_counter() - started
    print(f"{label}: {steps*fpb} frames in {took:.3f} s = {steps*fpb/took:.1f} frames/s = {steps*fpb*frame_bytes/took/1e9:.1f} GB/s of frames", flush=True)


run(device, "render into HBM")
run(host, "render into pinned host memory")
run(device, "render into HBM")
run(host, "render into pinned host memory")
import numpy as np  # noqa: E402
tape.build(0, fpb); tape.render(fpb, device[0].data_ptr()); tape.build(0, fpb); tape.render(fpb, host[0].data_ptr()); context.synchronize()
print("identical:", bool(np.array_equal(device[0].cpu().numpy(), host[0].numpy())))
