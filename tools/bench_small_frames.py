import sys, time
sys.path.insert(0, '/root/repo')
import examples.scenes as S
from shaderflow_amd import synth
S.make(S.Basic).main(width=256, height=256, fps=60, time=1.0, output="/dev/null")
for name, kw, w, h, ssaa, secs in (("Basic", {}, 256, 256, 1, 60.0), ("Basic", {}, 256, 256, 1, 60.0), ("MusicBars", dict(audio=(synth.sweep_clip(60.0, 44100), 44100)), 640, 360, 2, 60.0)):
    scene = S.make(getattr(S, name), **kw)
    t0 = time.perf_counter(); scene.main(width=w, height=h, ssaa=ssaa, fps=60, time=secs, output="/dev/null"); dt = time.perf_counter() - t0
    print(f"{name} {w}x{h} ssaa {ssaa}: {secs*60/dt:9.1f} frames/s")
