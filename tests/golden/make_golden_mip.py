#!/usr/bin/env python3
"""
Mipmapped textures on the reference (texture.py:116-137, 274-283: `mipmaps=True` → build_mipmaps() + LINEAR_MIPMAP_LINEAR) → mip.npz.
Everything is rendered by /root/reference's own Python through Mesa llvmpipe (refhost.py), as mesa.npz is.

levels.*   what glGenerateMipmap leaves in the levels of small textures (even, odd and 13x4 extents; unorm8 and float32), read back
           level by level: each level is the LINEAR-filtered, edge-clamped half-size image of the one above.
lod.*      the level of detail llvmpipe selects: a float32 texture whose level k holds the constant k/8, sampled at a sweep of
           texel-per-pixel ratios (axis-aligned and rotated) into a float32 target → the effective lambda, unrounded; and the same
           with unorm8 levels k·16 → the 8-bit blend between two levels.
probe.*    a fragment of this repository's own — texture(probe, R·astuv·S) — over unorm8 and float32 textures, "linear" and
           "nearest", in the two orders a user of the reference can end up with: `from_numpy(data)` alone (make() → apply() builds
           the chain BEFORE write() fills level 0: the chain is what glGenerateMipmap made of the empty texture — zeros) and
           `from_numpy(data).repeat(True)` (apply() again: the chain of the data).
visualizer.mip   the reference's visualizer.frag over a 480x270 background with `mipmaps = True` (2.8 texels per pixel).

Run:  python tests/golden/make_golden_mip.py        (needs /root/reference; ≈ 1 min)
"""
from __future__ import annotations

import ctypes as C
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT))

import refhost  # noqa: E402

refhost.install()

import make_golden_mesa as M  # noqa: E402
from refhost import GL  # noqa: E402
from shaderflow.scene import ShaderScene  # noqa: E402
from shaderflow.texture import ShaderTexture  # noqa: E402

from tests.helpers import mip_probe_texture, visualizer_inputs  # noqa: E402


def read_levels(context, texture, width: int, height: int, components: int, dtype) -> list[np.ndarray]:
    gl, levels, level = context.gl, [], 0
    while True:
        image = np.zeros((height, width, components), dtype)
        gl.glBindTexture(GL["TEXTURE_2D"], texture.glo)
        gl.glPixelStorei(0x0D05, 1)                                  # GL_PACK_ALIGNMENT
        gl.glGetTexImage(GL["TEXTURE_2D"], level, texture._base, texture._kind, C.c_void_p(image.ctypes.data))
        levels.append(image)
        if width == 1 and height == 1:
            return levels
        width, height, level = max(1, width//2), max(1, height//2), level + 1


def sampled(texels: np.ndarray, width: int, height: int, scale, rotation: float, *, filter="linear", reapply=True, constant_levels=None,
            target=np.uint8) -> np.ndarray:
    """texture(probe, R(rotation)·(astuv·scale)) over a width x height target → (height, width, 4), row 0 = bottom"""
    c, s = float(np.cos(rotation)), float(np.sin(rotation))
    fragment = (f"void main() {{ vec2 p = astuv*vec2({float(scale[0])!r}, {float(scale[1])!r}); "
                f"fragColor = texture(probe, vec2({c!r}*p.x - {s!r}*p.y, {s!r}*p.x + {c!r}*p.y)); }}")

    class Probe(ShaderScene):
        def build(self):
            texture = ShaderTexture(scene=self, name="probe", filter=filter, mipmaps=True)
            texture.from_numpy(np.flipud(texels))
            if reapply:
                texture.repeat(True)                                # apply(): build_mipmaps() from the data
            if constant_levels is not None:                           # the harness overwrites every level with a constant (lod.* only)
                raw, level, n = texture.get_box().texture, 0, texels.shape[0]
                while n >= 1:
                    raw.write(np.full((n, n, texels.shape[2]), constant_levels(level), texels.dtype), viewport=(0, 0, n, n), level=level)
                    level, n = level + 1, n//2
            self.shader.fragment = fragment
            self.shader.texture.dtype = target

    scene = Probe()
    refhost.export(scene, width=width, height=height, ssaa=1.0, subsample=1, fps=60.0, time=1/60, tag="probe")
    box = scene.shader.texture.get_box().texture
    return np.frombuffer(box.read(), target).reshape(box.size[1], box.size[0], 4).copy()


def main() -> None:
    out: dict[str, np.ndarray] = {}
    context = refhost.Context()
    out["meta.renderer"] = np.array(f"{context.info['GL_VERSION']} | {context.info['GL_RENDERER']}")
    rng = np.random.default_rng(3)
    # --- glGenerateMipmap, level by level ------------------------------------------------------------------------------------------
    for (w, h, dtype) in ((8, 6, np.uint8), (7, 5, np.uint8), (13, 4, np.uint8), (8, 6, np.float32), (7, 5, np.float32)):
        data = rng.integers(0, 256, (h, w, 4)).astype(np.uint8) if dtype == np.uint8 else rng.random((h, w, 4), dtype=np.float32)
        texture = refhost.Texture(context, (w, h), 4, data, dtype="f1" if dtype == np.uint8 else "f4")
        texture.build_mipmaps()
        tag = f"levels.{w}x{h}.{np.dtype(dtype).name}"
        for k, level in enumerate(read_levels(context, texture, w, h, 4, dtype)):
            out[f"{tag}.{k}"] = level
        print(tag, [out[f"{tag}.{k}"].shape[:2] for k in range(8) if f"{tag}.{k}" in out])
    # --- the level of detail, measured -----------------------------------------------------------------------------------------------
    W = 64
    sweep = [0.5, 0.9, 1.0, 1.05, 1.2, 1.375, 1.5, 1.9, 2.0, 2.1, 2.5, 3.0, 3.9, 4.0, 5.0, 7.0, 8.0, 11.0]
    zeros32, zeros8 = np.zeros((256, 256, 4), np.float32), np.zeros((256, 256, 4), np.uint8)
    measured = []
    for rho in sweep:
        image = sampled(zeros32, W, W, (rho*W/256.0, 0.5*rho*W/256.0), 0.0, constant_levels=lambda k: k/8.0, target=np.float32)
        assert np.ptp(image[..., 0])*8.0 < 2e-3, (rho, np.ptp(image[..., 0])*8.0)      # one lambda for the whole quad-aligned image
        measured.append(float(np.median(image[..., 0]))*8.0)
    out["lod.rho"], out["lod.lambda"] = np.array(sweep), np.array(measured)
    rotated = []
    for rotation, rho in ((0.3, 1.5), (0.3, 3.0), (np.pi/4, 1.5), (np.pi/4, 3.0)):
        image = sampled(zeros32, W, W, (rho*W/256.0, rho*W/256.0), rotation, constant_levels=lambda k: k/8.0, target=np.float32)
        rotated.append((rotation, rho, float(image[..., 0].min())*8.0, float(image[..., 0].max())*8.0))
    out["lod.rotated"] = np.array(rotated)
    blend = []
    for k in range(65):
        rho = 1.0 + k/64.0
        image = sampled(zeros8, W, W, (rho*W/256.0, 0.5*rho*W/256.0), 0.0, constant_levels=lambda level: level*16, target=np.float32)
        blend.append((rho, float(image[0, 0, 0])*255.0))
    out["lod.blend_u8"] = np.array(blend)
    print("lambda(rho):", [f"{r}:{v:.4f}" for r, v in zip(sweep, measured)])
    # --- textures of the parity tests, both build orders, both filters -------------------------------------------------------------
    for dtype in (np.uint8, np.float32):
        texels = mip_probe_texture(64, 48, dtype)
        name = np.dtype(dtype).name
        for tag, scale, rotation in (("magnified", (0.6, 0.5), 0.2), ("x1.6", (2.4, 2.0), 0.15), ("x3.3", (5.0, 4.1), -0.4), ("x9", (13.0, 11.0), 0.0)):
            out[f"probe.{name}.linear.{tag}"] = sampled(texels, 96, 54, scale, rotation)
        out[f"probe.{name}.nearest.x3.3"] = sampled(texels, 96, 54, (5.0, 4.1), -0.4, filter="nearest")
        out[f"probe.{name}.linear.x3.3.stale"] = sampled(texels, 96, 54, (5.0, 4.1), -0.4, reapply=False)
    # --- a fragment of the reference over a mipmapped background ------------------------------------------------------------------------
    u, arrays, params = visualizer_inputs(160, 90, seed=21, volume=0.5, bg_size=(480, 270))

    def mipmapped_background(scene):
        next(m for m in scene.modules if getattr(m, "name", None) == "background").mipmaps = True      # __apply__: the chain of the image

    screen, frame = M.probe(M.EXAMPLES/"visualizer.frag", 160, 90, textures=arrays, params=params, uniforms=M.oracle_inputs(u), configure=mipmapped_background)
    out["visualizer.mip.image"], out["visualizer.mip.final"] = screen, frame
    plain, _ = M.probe(M.EXAMPLES/"visualizer.frag", 160, 90, textures=arrays, params=params, uniforms=M.oracle_inputs(u))
    d = np.abs(screen.astype(int) - plain.astype(int))
    print(f"visualizer.mip: {100*(d > 1).mean():.1f} % of the values differ from the unmipmapped image by more than 1 LSB (max {d.max()})")
    assert (d > 1).mean() > 0.2
    np.savez_compressed(HERE/"mip.npz", **out)
    print(f"wrote {HERE/'mip.npz'} ({(HERE/'mip.npz').stat().st_size >> 10} KiB)")


if __name__ == "__main__":
    main()
