#!/usr/bin/env python3
"""
How far is the kernel that produces the benchmark number from the reference's GLSL?  

For the visualizer images of tests/golden/gles.npz (the reference's visualizer.frag / final.glsl rendered by an independent OpenGL
implementation, SwiftShader) this prints, per image: the histogram of |HIP - GLSL| per channel value, the same for the parity
oracle, and for every value that is 2 LSB off its location, what the oracle says there and whether the pixel sits on a decision
edge of the fragment (a supersample next to the bar outline / a spectrogram bin boundary / the waveform strip), i.e. where two
conforming GL implementations legitimately pick different sides. Runs on the GPU box:  python tools/parity_histogram.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT/"tests"))

from oracle import binding as O                                   # noqa: E402
from tests.helpers import Gpu, gpu_bind_all, i16_to_f32, oracle_textures, visualizer_inputs   # noqa: E402

G = np.load(ROOT/"tests"/"golden"/"gles.npz")


def histogram(tag: str, got: np.ndarray, want: np.ndarray) -> np.ndarray:
    d = np.abs(got.astype(int) - want.astype(int))
    counts = np.bincount(d.ravel(), minlength=4)
    total = d.size
    print(f"  {tag:34s} " + "  ".join(f"|d|={k}: {counts[k]:8d} ({100*counts[k]/total:7.4f} %)" for k in range(min(len(counts), 4)))
          + (f"  max {d.max()}" if d.max() > 3 else ""))
    return d


def edge_mask(image: np.ndarray) -> np.ndarray:
    """pixels whose 3x3 neighbourhood spans more than 24 LSB in some channel: outlines of the bars, the strips, bin boundaries"""
    pad = np.pad(image.astype(int), ((1, 1), (1, 1), (0, 0)), mode="edge")
    lo = np.full(image.shape, 255); hi = np.zeros(image.shape, int)
    for dy in range(3):
        for dx in range(3):
            window = pad[dy:dy + image.shape[0], dx:dx + image.shape[1]]
            lo = np.minimum(lo, window); hi = np.maximum(hi, window)
    return ((hi - lo) > 24).any(axis=2)


def main() -> None:
    gpu = Gpu()
    print("# |HIP - reference GLSL (SwiftShader)| per channel value; kernel named by sfx_last_kernel()\n")
    print("## visualizer.frag alone, 160x90, white-noise 120x68 background, three volumes (tests/test_gpu_gles.py::test_visualizer_tiled_kernel)")
    for volume in (0.0, 0.5, 1.2):
        u, arrays, params = visualizer_inputs(160, 90, seed=21, volume=volume, bg_size=(120, 68))
        prog, _ = gpu.program("visualizer")
        gpu.set_uniforms(prog, u)
        gpu_bind_all(gpu, prog, arrays, params)
        got = gpu.render(prog, 160, 90)
        kernel = gpu.lib.sfx_last_kernel().decode()
        want = G[f"visualizer.v{volume}.image"]
        oracle = O.render("visualizer", u, oracle_textures(arrays, params), 160, 90, threads=8)
        print(f" volume {volume}: {kernel}")
        d = histogram("HIP vs GLSL", got, want)
        histogram("oracle vs GLSL", oracle, want)
        histogram("HIP vs oracle", got, oracle)
        report_twos(d, got, want, oracle)

    print("\n## the benchmark's configuration: bands of the 3840x2160 2xSSAA frame (tests/golden/gles_4k.npz; test_benchmark_kernel_against_the_reference_glsl_at_4k)")
    K = np.load(ROOT/"tests"/"golden"/"gles_4k.npz")
    w, h, ssaa, seed, volume = int(K["args"][0]), int(K["args"][1]), int(K["args"][2]), int(K["args"][3]), float(K["args"][4])
    u, arrays, params = visualizer_inputs(w, h, seed=seed, volume=volume, bg_size=(int(K["args"][5]), int(K["args"][6])))
    u.iSSAA = float(ssaa)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    frame = gpu.render_resolve(prog, w, h, ssaa, 2)
    print(f" {gpu.lib.sfx_last_kernel().decode()}")
    for first, last in K["bands"]:
        want = K[f"rows{first}.final"]
        screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, rows=(first*ssaa, last*ssaa), threads=8)
        oracle = O.resolve(screen, w, h, 2, rows=(first, last), threads=8)[first:last]
        print(f"  output rows {first}-{last}:")
        d = histogram("HIP vs GLSL", frame[first:last], want)
        histogram("oracle vs GLSL", oracle, want)
        histogram("HIP vs oracle", frame[first:last], oracle)
        report_twos(d, frame[first:last], want, oracle)

    print("\n## whole export, Visualizer scene at 2x SSAA through the frame tape (test_end_to_end_export_against_the_reference_pipeline)")
    from examples.scenes import Visualizer, make
    from shaderflow_amd import synth
    P = np.load(ROOT/"tests"/"golden"/"pipeline.npz")
    fps, samplerate, frames = float(P["meta"][0]), int(P["meta"][1]), int(P["meta"][2])
    w, h, ssaa = (int(v) for v in G["frames.size"])
    for batch in (None, False):
        scene = make(Visualizer, audio=(i16_to_f32(P["pcm_i16"]), samplerate), background=synth.background_image(240, 135, seed=7))
        raw = scene.main(width=w, height=h, fps=fps, ssaa=ssaa, subsample=2, time=frames/fps, output=bytes, batch=batch)
        got = np.frombuffer(raw, np.uint8).reshape(-1, h, w, 3)
        print(f" {'frame tape' if batch is None else 'frame loop'} ({w}x{h}, ssaa {ssaa}): {gpu.lib.sfx_last_kernel().decode()}")
        for k in G["frames.index"]:
            want = G[f"frames.{k}"]
            d = histogram(f"frame {int(k)} HIP vs GLSL", got[k], want)
            report_twos(d, got[k], want, None)


def report_twos(d, got, want, oracle) -> None:
    where = np.argwhere(d >= 2)
    if not len(where):
        return
    edges = edge_mask(want)
    on_edge = sum(bool(edges[y, x]) for y, x, _ in where)
    print(f"      {len(where)} values at >= 2 LSB, {on_edge} of them on an outline of the reference image (3x3 range > 24 LSB):")
    for y, x, c in where[:24]:
        line = f"        (x {x:3d}, y {y:3d}, channel {c}): HIP {got[y, x, c]:3d}  GLSL {want[y, x, c]:3d}"
        if oracle is not None:
            line += f"  oracle {oracle[y, x, c]:3d}"
        print(line + ("  [outline]" if edges[y, x] else ""))
    if len(where) > 24:
        print(f"        … {len(where) - 24} more")


if __name__ == "__main__":
    main()
