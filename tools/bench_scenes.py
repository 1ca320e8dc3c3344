"""Whole exports (scene.main → /dev/null, read-out included) of the example scenes at 1080p, 600 frames each; the python frame loop
(scenes with logic between frames, layered/temporal textures) and the frame tape side by side."""
import sys, time
sys.path.insert(0, '/root/repo')
import examples.scenes as S
from shaderflow_amd import synth
from shaderflow_amd.tape import FrameTape

audio = (synth.sweep_clip(20.0, 44100), 44100)
background = synth.background_image(1920, 1080, seed=0)
cases = [("Basic", {}, 2), ("Basic", {}, 2), ("Waveform", dict(audio=audio), 2), ("MusicBars", dict(audio=audio), 2), ("Visualizer", dict(audio=audio, background=background), 2),
         ("Visualizer", dict(audio=audio, background=background), 1), ("Dynamics", {}, 2), ("MultiShader", {}, 2), ("Multipass", {}, 1), ("MotionBlur", {}, 1), ("Life", {}, 1),
         ("RayMarch", {}, 2), ("Mandelbrot", {}, 2), ("ShaderToy", {}, 2), ("Plasma", {}, 2), ("Bloom", dict(background=background), 2),
         ("Bloom no tile", dict(background=background), 2)]
import os
for k, (name, kw, ssaa) in enumerate(cases):
    os.environ["SHADERFLOW_JIT_TILE"] = "0" if name.endswith("no tile") else "1"      # Bloom: its translated fragment with / without the LDS tile
    label, name = name, name.split()[0]
    if name in ("Plasma", "Bloom"):                            # fragments of their own: compile (hipcc, seconds) outside the timing, the cache serves the run
        S.make(getattr(S, name), **kw).main(width=1920, height=1080, ssaa=ssaa, fps=60, time=2/60, output="/dev/null")
    scene = S.make(getattr(S, name), **kw)
    t0 = time.perf_counter()
    scene.main(width=1920, height=1080, ssaa=ssaa, fps=60, time=10.0, output="/dev/null")
    dt = time.perf_counter() - t0
    probe = S.make(getattr(S, name), **kw); probe.initialize()
    print(f"{label:13s} ssaa {ssaa}  {'tape' if FrameTape.applicable(probe) else 'loop'}  600 frames 1080p: {600/dt:8.1f} frames/s" + ("   (process warm-up)" if k == 0 else ""))
