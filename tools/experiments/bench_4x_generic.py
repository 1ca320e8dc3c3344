"""Generic fused kernels at 4xSSAA (render_resolve_body<PlainShader/JitShader, 4>): frames/s of whole exports to /dev/null at 1080p"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import examples.scenes as S
from shaderflow_amd import synth

background = synth.background_image(1920, 1080, seed=0)
for name, kw in (("RayMarch", {}), ("Mandelbrot", {}), ("ShaderToy", {}), ("Plasma", {}), ("Basic", {})):
    S.make(getattr(S, name), **kw).main(width=1920, height=1080, ssaa=4, fps=60, time=2/60, output="/dev/null")   # compile / warm up
    best = 0.0
    for _ in range(2):
        scene = S.make(getattr(S, name), **kw)
        t0 = time.perf_counter()
        scene.main(width=1920, height=1080, ssaa=4, fps=60, time=4.0, output="/dev/null")
        best = max(best, 240/(time.perf_counter() - t0))
    print(f"{name:12s} 1080p 4xSSAA: {best:8.1f} frames/s", flush=True)
