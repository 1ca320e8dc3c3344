#!/usr/bin/env python3
"""
Golden images for run-time translated fragments on DESKTOP OpenGL: the fragments of tests/golden/jit/*.glsl — written for this
repository — set as `scene.shader.fragment` of a scene of THE REFERENCE (its own ShaderScene / ShaderProgram assemble the source:
`#version 330`, its prelude, its typed uniforms) and rendered by Mesa llvmpipe (refhost.py, make_golden_mesa.py's `probe`). Uniforms
the fragments declare themselves are set the way a user of the reference would, with `shader.set_uniform(name, value)` from a module's
update(). Same cases and keys as jit.npz (SwiftShader, GLSL ES) → jit_mesa.npz; tests/test_gpu_translated.py holds the translated
code objects to both.

usage: python tests/golden/make_golden_jit_mesa.py
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))

import make_golden_mesa as M  # noqa: E402  (installs the reference host)
from attrs import define  # noqa: E402
from make_golden_jit import CASES, FRAGMENTS, background  # noqa: E402
from oracle import binding as O  # noqa: E402
from shaderflow.module import ShaderModule  # noqa: E402


@define
class UserUniforms(ShaderModule):
    """What a scene of the reference does for uniforms its fragment declares itself: program[name].value = … before every frame"""
    values: dict = None

    def update(self):
        for name, value in (self.values or {}).items():
            self.scene.shader.set_uniform(name, value)


def main() -> None:
    out: dict[str, np.ndarray] = {"background": background()}
    for name, (w, h, overrides, floats, integers, _) in CASES.items():
        if w % 2 or h % 2:
            print(f"{name:18s} {w}x{h}: skipped — scene.main() fits resolutions to even numbers (resolution.py:6-86); SwiftShader's image stays the witness")
            continue
        u = O.default_uniforms(w, h, **overrides)
        text = (FRAGMENTS/f"{name.split('.')[0]}.glsl").read_text()
        own = {**{k: (tuple(v) if hasattr(v, "__len__") else float(v)) for k, v in floats.items()}, **{k: int(v) for k, v in integers.items()}}
        screen, _ = M.probe(text, w, h, textures={"background": out["background"]}, params={"background": ("linear", True, True)},
                            uniforms=M.oracle_inputs(u), configure=lambda scene, own=own: UserUniforms(scene=scene, name="user", values=own))
        out[f"{name}.image"] = screen
        print(f"{name:18s} {w}x{h} mean {screen[..., :3].mean():6.1f}")
    out["cases"] = np.array(json.dumps({name: dict(width=w, height=h, uniforms=overrides, floats=floats, integers=integers)
                                        for name, (w, h, overrides, floats, integers, _) in CASES.items() if f"{name}.image" in out}))
    np.savez_compressed(HERE/"jit_mesa.npz", **out)
    print("wrote", HERE/"jit_mesa.npz", (HERE/"jit_mesa.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
