"""
Test infrastructure: a translated fragment (shaderflow_amd/glsl2hip.py) built for the HOST — the same translation unit and
the same header (csrc/jit_runtime.hpp, which is `__host__ __device__` throughout) compiled with -DSF_JIT_HOST into a
shared library whose `sfx_jit_host_render` shades every pixel on the CPU. Lets the CPU suite compare translations with
the parity oracle and with the OpenGL goldens without a GPU. No product path uses this.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import subprocess
from pathlib import Path

import numpy as np

from oracle import binding as O
from shaderflow_amd import glsl2hip as G

_DTYPES = {np.dtype(np.uint8): 0, np.dtype(np.float32): 1, np.dtype(np.uint16): 2, np.dtype(np.float16): 3}      # csrc/glsl.hpp DT_*


class HostFragment:
    def __init__(self, translation: G.Translation, directory: Path):
        self.translation = translation
        directory.mkdir(parents=True, exist_ok=True)
        key = hashlib.sha256((translation.cpp + G.runtime_fingerprint()).encode()).hexdigest()[:20]
        unit, library = directory/f"host_{key}.hip", directory/f"host_{key}.so"
        if not library.exists():
            unit.write_text(translation.cpp)
            done = subprocess.run([G.HIPCC, *G.FLAGS, "-DSF_JIT_HOST", "-shared", "-fPIC", f"-I{G.CSRC}", str(unit), "-o", str(library)],
                                  capture_output=True, text=True, timeout=600)
            assert done.returncode == 0, done.stderr[-4000:]
        self.lib = C.CDLL(str(library))
        self.lib.sfx_jit_host_uniforms_size.restype = C.c_size_t
        self.lib.sfx_jit_host_textures_size.restype = C.c_size_t
        self.uniforms = C.create_string_buffer(self.lib.sfx_jit_host_uniforms_size())
        self.textures = C.create_string_buffer(self.lib.sfx_jit_host_textures_size())
        self.lib.sfx_jit_host_defaults(self.uniforms)
        self._keep: list[np.ndarray] = []
        for binding in translation.bindings:                 # what ShaderProgram._load_translated does with `uniform T x = …;`
            if binding.default is not None:
                self.set(binding.name, binding.default)

    def set_uniforms(self, u: O.Uniforms) -> None:
        """every built-in field of an oracle uniform block, by name"""
        for name, _ in u._fields_:
            if name == "user":
                continue
            value = getattr(u, name)
            values = np.array(list(value) if hasattr(value, "__len__") else [value], np.float32)
            assert self.lib.sfx_jit_host_uniform(self.uniforms, name.encode(), values.ctypes.data_as(C.c_void_p), len(values)), name

    def set(self, name: str, value) -> None:
        """a scene-defined uniform, through the translation's bindings"""
        elements = [b for b in self.translation.bindings if b.array == name]
        if elements:                                         # a uniform array: element by element (ShaderProgram._push)
            for index, row in enumerate(np.asarray(value, np.float64).reshape(len(elements), -1)):
                self.set(f"{name}[{index}]", row)
            return
        binding = next(b for b in self.translation.bindings if b.name == name and not b.sampler)
        words = np.atleast_1d(np.asarray(value, np.int32 if binding.integer else np.float32))
        assert words.size == binding.count, (name, words.size, binding.count)
        self.lib.sfx_jit_host_user(self.uniforms, binding.slot, words.ctypes.data_as(C.c_void_p), binding.count)

    def bind(self, name: str, data: np.ndarray, filter: str = "linear", repeat_x: bool = True, repeat_y: bool = True) -> bool:
        binding = next((b for b in self.translation.bindings if b.name == name and b.sampler), None)
        if binding is None:
            return False
        data = np.ascontiguousarray(data if data.ndim == 3 else data[:, :, None])
        self._keep.append(data)
        self.lib.sfx_jit_host_texture(self.textures, binding.slot, data.ctypes.data_as(C.c_void_p), data.shape[1], data.shape[0], data.shape[2],
                                      _DTYPES[data.dtype], 1 if filter == "linear" else 0, int(repeat_x), int(repeat_y))
        return True

    def render(self, width: int, height: int) -> np.ndarray:
        out = np.zeros((height, width, 4), np.uint8)
        self.lib.sfx_jit_host_render(self.uniforms, self.textures, width, height, out.ctypes.data_as(C.c_void_p))
        return out

    def render_float(self, width: int, height: int) -> np.ndarray:
        """the fragment's output before any target conversion: (h, w, 4) float32"""
        out = np.zeros((height, width, 4), np.float32)
        self.lib.sfx_jit_host_render_float(self.uniforms, self.textures, width, height, out.ctypes.data_as(C.c_void_p))
        return out
