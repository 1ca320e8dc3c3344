#!/usr/bin/env python3
"""
How far is the HIP path from THE REFERENCE ITSELF?  Histograms of |HIP - reference| per channel value against the frames that
/root/reference's own Python rendered on Mesa llvmpipe (tests/golden/mesa.npz, mesa_4k.npz), and of |HIP - oracle| on the same
inputs: kernels alone, the example scenes exported end to end, whole 3840x2160 2xSSAA frames. Runs on the GPU box:
    python tools/parity_histogram_r03.py > gpurun_out/r03/parity_histogram.txt
"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import binding as O                                   # noqa: E402
from shaderflow_amd import synth                                  # noqa: E402
from tests.helpers import Gpu, gpu_bind_all, i16_to_f32, oracle_textures, visualizer_inputs   # noqa: E402
from tests.test_oracle_mesa import c3_inputs                      # noqa: E402

G = np.load(ROOT/"tests"/"golden"/"mesa.npz")
THREADS = min(os.cpu_count() or 8, 16)


def histogram(tag: str, got: np.ndarray, want: np.ndarray) -> None:
    d = np.abs(got.astype(int) - want[..., :got.shape[-1]].astype(int))
    counts = np.bincount(np.minimum(d.ravel(), 4), minlength=5)
    print(f"  {tag:46s} " + "  ".join(f"|d|={k}{'+' if k == 4 else ''}: {counts[k]:9d} ({100*counts[k]/d.size:8.5f} %)" for k in range(5)) + f"  max {d.max()}")


def main() -> None:
    gpu = Gpu()
    print(f"# |HIP - reference on {G['meta.renderer']}| and |HIP - oracle| per channel value; kernels named by sfx_last_kernel()\n")
    print("## visualizer.frag alone, 160x90, white-noise 120x68 background (the worst case for a filter's weight precision)")
    for volume in (0.0, 0.5, 1.2):
        u, arrays, params = visualizer_inputs(160, 90, seed=21, volume=volume, bg_size=(120, 68))
        prog, _ = gpu.program("visualizer")
        gpu.set_uniforms(prog, u)
        gpu_bind_all(gpu, prog, arrays, params)
        got = gpu.render(prog, 160, 90)
        print(f" volume {volume}: {gpu.lib.sfx_last_kernel().decode()}")
        histogram("HIP vs reference (llvmpipe)", got, G[f"visualizer.v{volume}.image"])
        histogram("HIP vs oracle", got, O.render("visualizer", u, oracle_textures(arrays, params), 160, 90, threads=8))
        histogram("oracle vs reference", O.render("visualizer", u, oracle_textures(arrays, params), 160, 90, threads=8), G[f"visualizer.v{volume}.image"])

    print("\n## example scenes exported end to end (scene.main, frame tape), frames the fixture keeps")
    import examples.scenes as S
    P = np.load(ROOT/"tests"/"golden"/"pipeline.npz")
    audio = (i16_to_f32(P["pcm_i16"]), int(P["meta"][1]))
    background = synth.background_image(240, 135, seed=7)
    builders = {"basic": lambda: S.Basic(), "shadertoy": lambda: S.ShaderToy(), "raymarch": lambda: S.RayMarch(), "multishader": lambda: S.MultiShader(),
                "multipass": lambda: S.Multipass(), "motionblur": lambda: S.MotionBlur(), "dynamics": lambda: S.Dynamics(),
                "visualizer": lambda: S.make(S.Visualizer, audio=audio, background=background),
                "visualizer.ssaa1": lambda: S.make(S.Visualizer, audio=audio, background=background),
                "musicbars": lambda: S.make(S.MusicBars, audio=audio), "waveform": lambda: S.make(S.Waveform, audio=audio)}
    for tag, build in builders.items():
        width, height, ssaa, subsample, fps, frames = G[f"scene.{tag}.args"]
        raw = build().main(width=int(width), height=int(height), ssaa=(int(ssaa) if ssaa == int(ssaa) else float(ssaa)), subsample=int(subsample),
                           fps=float(fps), time=int(frames)/float(fps), output=bytes)
        got = np.frombuffer(raw, np.uint8).reshape(-1, int(height), int(width), 3)[G[f"scene.{tag}.index"]]
        histogram(f"{tag} {int(width)}x{int(height)} ssaa {ssaa:g}: HIP vs reference", got, G[f"scene.{tag}.frames"])

    print("\n## BASELINE config 3: whole 3840x2160 2xSSAA frames (reference: every 13th / 27th row of the frame it exported; oracle: the whole frame)")
    for name in ("noise", "bench"):
        K, u, arrays, params, w, h, ssaa = c3_inputs(name)
        prog, _ = gpu.program("visualizer")
        gpu.set_uniforms(prog, u)
        gpu_bind_all(gpu, prog, arrays, params)
        frame = gpu.render_resolve(prog, w, h, ssaa, 2)
        print(f" {name} background: {gpu.lib.sfx_last_kernel().decode()}")
        histogram(f"HIP vs reference, {len(K[f'{name}.rows'])} rows x 3840 px", frame[K[f"{name}.rows"]], K[f"{name}.final"])
        screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, threads=THREADS)
        oracle = O.resolve(screen, w, h, 2, threads=THREADS)
        histogram("HIP vs oracle, whole frame", frame, oracle)
        histogram(f"oracle vs reference, {len(K[f'{name}.rows'])} rows", oracle[K[f"{name}.rows"]], K[f"{name}.final"])
    gpu.close()


if __name__ == "__main__":
    main()
