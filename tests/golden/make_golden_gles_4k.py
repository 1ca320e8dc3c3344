#!/usr/bin/env python3
"""
The benchmark's own configuration against the reference's GLSL: bands of a 3840x2160 frame at 2x SSAA.

tests/golden/gles.npz holds whole images at sizes of ~160x90; the kernel the benchmark runs (k_visualizer_fast: per-frame
column/row tables, axis lines, 72x10-cell LDS tile) is only selected when a 128-pixel block's window of the background fits
its tile — at 4K over a 1080-row background, not at those sizes. So this renders, with the same assembly of the reference's
GLSL as make_golden_gles.py (visualizer.frag at 7680x4320, then final.glsl at 3840x2160; SwiftShader, OpenGL ES 3.0), three
BANDS of four output rows of the frame test_gpu_pixels.py::test_full_size_properties_4k_ssaa2 renders (scissor test: only the
band is shaded; the varyings are those of the full-screen quad), and stores them in gles_4k.npz (138 KB). No GLSL text is
stored. Inputs are regenerated from the seed by tests/helpers.py::visualizer_inputs.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT))

from gles import Context  # noqa: E402
from make_golden_gles import EXAMPLES, QUAD, SHADERS, build, uniform_values  # noqa: E402
from oracle import binding as O  # noqa: E402
from tests.helpers import visualizer_inputs  # noqa: E402

W, H, SSAA, SEED, VOLUME = 3840, 2160, 2, 51, 0.9
BANDS = ((0, 4), (1000, 1004), (2156, 2160))                   # output rows [first, last)


def main() -> None:
    ctx = Context()
    print(ctx.version)
    u, arrays, params = visualizer_inputs(W, H, seed=SEED, volume=VOLUME, bg_size=(1920, 1080))
    u.iSSAA = float(SSAA)
    vertex, fragment = build((EXAMPLES/"visualizer.frag").read_text(), ["background", "iSpectrogram", "iWaveform"])
    visualizer = ctx.program(vertex, fragment)
    vertex, fragment = build((SHADERS/"fragment/final.glsl").read_text().replace("uniform int iSubsample;", ""), ["iScreen"])
    final = ctx.program(vertex, fragment)
    as_bool = lambda p: (p[0] == "linear", bool(p[1]), bool(p[2]))
    handles = {name: ctx.texture(data, *as_bool(params[name])) for name, data in arrays.items()}
    out = {"args": np.array([W, H, SSAA, SEED, VOLUME, 1920, 1080], np.float64), "bands": np.array(BANDS)}
    wr, hr = W*SSAA, H*SSAA
    for first, last in BANDS:
        band = ctx.draw(visualizer, wr, hr, uniform_values(u), handles, {"vertex_position": QUAD, "vertex_gluv": QUAD},
                        region=(0, first*SSAA, wr, (last - first)*SSAA))
        screen = np.zeros((hr, wr, 4), np.uint8)
        screen[first*SSAA:last*SSAA] = band
        frame = ctx.draw(final, W, H, uniform_values(O.default_uniforms(W, H), iSubsample=2), {"iScreen": ctx.texture(screen, True, False, False)},
                         {"vertex_position": QUAD, "vertex_gluv": QUAD}, region=(0, first, W, last - first))
        out[f"rows{first}.final"] = frame[..., :3].copy()
        out[f"rows{first}.screen"] = band[:, ::16].copy()          # every 16th supersample column of iScreen: the unfused pass has its witness too
        print(f"rows {first}-{last}: final mean {frame[..., :3].mean():.2f}")
    np.savez_compressed(HERE/"gles_4k.npz", **out)
    print("gles_4k.npz", (HERE/"gles_4k.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
