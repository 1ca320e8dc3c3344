#!/usr/bin/env python3
"""
Two facts about the implementation the mesa goldens were rendered on, as DATA → tests/golden/filter.npz.

filter.*   How Mesa llvmpipe filters unorm8 textures. A fragment of this repository's own — `fragColor = texture(probe, astuv*S + O)` —
           as the main shader of a scene of the reference, rendered INTO A FLOAT32 TARGET so that the filtered values arrive unrounded:
           a 4-texel row swept in 4096 steps (one texel = 1024 steps = 4 steps per 1/256), and a 10x6 random grid at a skewed scale,
           repeat and clamp. Every value turns out to be k/255 exactly (stored: k as uint8, and the largest distance from k/255 seen);
           tests/test_oracle_mesa.py checks that the oracle's llvmpipe switch reproduces all of them.
aniso.*    What `texture.anisotropy = 16` (texture.py:280) does on llvmpipe when the context exposes EXT_texture_filter_anisotropic
           (`REFHOST_ANISOTROPY=1`; the mesa.npz goldens are rendered with the extension masked, refhost.py Context): the probes
           `default.plain` and `visualizer.v0.5` of mesa.npz once more, same inputs, extension on. Stored with the measured deviation
           from the isotropic image, so the decision to pin parity on the OpenGL 3.3 core filter is a fixture, not a paragraph.

Run:  python tests/golden/make_golden_filter.py      (needs /root/reference; ≈ 1 min)
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT))

ANISO = os.environ.get("REFHOST_ANISOTROPY") == "1"

import refhost  # noqa: E402

refhost.install()

import make_golden_mesa as M  # noqa: E402
from shaderflow.scene import ShaderScene  # noqa: E402
from shaderflow.texture import ShaderTexture  # noqa: E402

from tests.helpers import visualizer_inputs  # noqa: E402


def filtered(texels: np.ndarray, width: int, height: int, scale, offset, repeat: bool) -> np.ndarray:
    """texture(probe, astuv*scale + offset) over a width x height target of float32 → (height, width, 4) float32, row 0 = bottom"""
    fragment = f"void main() {{ fragColor = texture(probe, astuv*vec2({scale[0]!r}, {scale[1]!r}) + vec2({offset[0]!r}, {offset[1]!r})); }}"

    class Probe(ShaderScene):
        def build(self):
            texture = ShaderTexture(scene=self, name="probe", filter="linear", repeat_x=repeat, repeat_y=repeat)
            texture.from_numpy(np.flipud(texels))
            self.shader.fragment = fragment
            self.shader.texture.dtype = np.float32

    scene = Probe()
    refhost.export(scene, width=width, height=height, ssaa=1.0, subsample=1, fps=60.0, time=1/60, tag="probe")
    box = scene.shader.texture.get_box().texture
    return np.frombuffer(box.read(), np.float32).reshape(box.size[1], box.size[0], 4).copy()


def anisotropic_probes() -> dict:
    out = {}
    screen, frame = M.probe(M.SHADERS/"fragment/default.glsl", 160, 90, uniforms=dict(iTime=0.75, iTau=0.3))
    out["aniso.default.plain.image"], out["aniso.default.plain.final"] = screen, frame
    u, arrays, params = visualizer_inputs(160, 90, seed=21, volume=0.5, bg_size=(120, 68))
    screen, frame = M.probe(M.EXAMPLES/"visualizer.frag", 160, 90, textures=arrays, params=params, uniforms=M.oracle_inputs(u))
    out["aniso.visualizer.v0.5.image"], out["aniso.visualizer.v0.5.final"] = screen, frame
    return out


def main() -> None:
    if ANISO:                                                         # child: the context with the extension exposed
        context = refhost.Context()
        assert context.max_anisotropy > 1.0, "this Mesa does not expose EXT_texture_filter_anisotropic"
        out = anisotropic_probes()
        out["aniso.max"] = np.array(context.max_anisotropy)
        np.savez_compressed(sys.argv[1], **out)
        return
    out: dict[str, np.ndarray] = {}
    context = refhost.Context()
    out["meta.renderer"] = np.array(f"{context.info['GL_VERSION']} | {context.info['GL_RENDERER']}")
    rng = np.random.default_rng(5)
    row = np.zeros((1, 4, 4), np.uint8)
    row[0, :, 0], row[0, :, 1], row[0, :, 2], row[0, :, 3] = [0, 255, 0, 100], [10, 20, 200, 37], [255, 0, 255, 1], 255
    grid = rng.integers(0, 256, (6, 10, 4), dtype=np.uint8)
    out["filter.row.texels"], out["filter.grid.texels"] = row, grid
    deviation = 0.0
    for tag, texels, size, scale, offset, repeat in (("row", row, (4096, 2), (1.0, 1.0), (0.0, 0.0), True),
                                                     ("grid.repeat", grid, (320, 224), (1.7, 1.9), (-0.35, -0.45), True),
                                                     ("grid.clamp", grid, (320, 224), (1.7, 1.9), (-0.35, -0.45), False)):
        values = filtered(texels, *size, scale, offset, repeat)*np.float32(255.0)
        nearest = np.rint(values)
        deviation = max(deviation, float(np.abs(values - nearest).max()))
        out[f"filter.{tag}.k"] = nearest.astype(np.uint8)
        out[f"filter.{tag}.args"] = np.array([*size, *scale, *offset, int(repeat)], np.float64)
        print(f"filter.{tag:12s} {size[0]}x{size[1]}: largest |255·value − k| = {np.abs(values - nearest).max():.2e}")
    out["filter.max_distance_from_k_over_255"] = np.array(deviation/255.0)
    # the anisotropic context is a process of its own (the extension switch is read when refhost is imported)
    scratch = refhost.WORK/"aniso.npz"
    subprocess.run([sys.executable, str(Path(__file__).resolve()), str(scratch)], check=True, env=dict(os.environ, REFHOST_ANISOTROPY="1"))
    with np.load(scratch) as A:
        G = np.load(HERE/"mesa.npz")
        for name in A.files:
            out[name] = A[name]
        for tag in ("default.plain", "visualizer.v0.5"):
            for key in ("image", "final"):
                d = np.abs(A[f"aniso.{tag}.{key}"].astype(int) - G[f"{tag}.{key}"].astype(int))
                out[f"aniso.{tag}.{key}.deviation"] = np.array([float((d > 0).mean()), float((d > 1).mean()), float(d.max())])
                print(f"aniso.{tag}.{key}: {100*(d > 0).mean():.1f} % of the values differ from the isotropic one, {100*(d > 1).mean():.1f} % by more than 1 LSB, max {d.max()}")
    np.savez_compressed(HERE/"filter.npz", **out)
    print(f"wrote {HERE/'filter.npz'} ({(HERE/'filter.npz').stat().st_size >> 10} KiB)")


if __name__ == "__main__":
    main()
