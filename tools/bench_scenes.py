"""Whole exports (scene.main → /dev/null, read-out included) of a few example scenes at 1080p 2xSSAA, 600 frames each."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from examples.scenes import Waveform, MusicBars, Basic, make
from shaderflow_amd import synth
for cls in (Basic, Waveform, MusicBars, Basic):        # the first one pays the process warm-up
    kw = dict(audio=(synth.sweep_clip(20.0, 44100), 44100)) if cls is not Basic else {}
    scene = make(cls, **kw)
    t0 = time.perf_counter()
    scene.main(width=1920, height=1080, ssaa=2, fps=60, time=10.0, output="/dev/null")
    dt = time.perf_counter() - t0
    print(cls.__name__, "600 frames 1080p 2xSSAA:", round(600/dt, 1), "frames/s")
