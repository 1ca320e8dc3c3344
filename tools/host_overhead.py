import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from examples.scenes import MusicBars, make
from shaderflow_amd import _native as N, synth
from shaderflow_amd.tape import FrameTape
from shaderflow_amd.message import ShaderMessage
ctx = N.Context(0, torch.cuda.current_stream().cuda_stream)
pcm = synth.sweep_clip(20.0, 44100)
scene = make(MusicBars, audio=(pcm, 44100), context=ctx)
scene.initialize(); scene.exporting = scene.freewheel = scene.headless = True; scene.realtime = False
scene.relay(ShaderMessage.Shader.Compile); scene.resize(width=1920, height=1080)
for m in scene.modules: m.setup()
scene.set_duration(20.0); scene.ssaa = 1
tape = FrameTape(scene, batch=60).prepare(600); tape.bind_static_uniforms()
buf = torch.empty(60*1920*1080*3, dtype=torch.uint8, device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tape.build(i*60, 60); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    tape.render(60, buf.data_ptr()); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"build call {1e3*(t1-t0):.2f} ms, build gpu-drain {1e3*(t2-t1):.2f}, render call {1e3*(t3-t2):.2f}, render gpu-drain {1e3*(t4-t3):.2f}")
