"""
Device-side texture objects: what the reference gets from moderngl (`Texture` + `Framebuffer`) is one handle of
libshaderflow_hip here (`sfx_texture_*`, include/shaderflow_hip.h).
"""
from __future__ import annotations

import ctypes as C
import itertools
from typing import Optional

import numpy as np
from attrs import define, field

from shaderflow_amd import _native as N


@define(eq=False, slots=False)
class DeviceTexture:
    """What a sampler uniform carries: one native texture (stands for moderngl.Texture)"""
    context: N.Context
    handle: N.Handle
    size: tuple[int, int]
    components: int
    dtype: np.dtype
    serial: int = field(factory=itertools.count(1).__next__)
    """Unique per allocation (a handle value can be reused after a release): what sampler-binding caches compare"""

    @property
    def nbytes(self) -> int:
        return self.size[0]*self.size[1]*self.components*self.dtype.itemsize

    def params(self, filter: str, repeat_x: bool, repeat_y: bool, mipmaps: bool = False) -> None:
        """`mipmaps`: the minification filter becomes LINEAR_MIPMAP_LINEAR / NEAREST_MIPMAP_NEAREST (moderngl_filter, texture.py:131-137)"""
        linear = (filter == "linear")
        code = (2 if linear else 3) if mipmaps else (N.LINEAR if linear else N.NEAREST)      # SFX_LINEAR_MIPMAP_LINEAR / SFX_NEAREST_MIPMAP_NEAREST
        N.check(N.lib().sfx_texture_params(self.handle, code, int(repeat_x), int(repeat_y)))

    def build_mipmaps(self) -> None:
        """moderngl.Texture.build_mipmaps(): levels 1… from the current level 0 (later writes touch level 0 only, as in OpenGL)"""
        N.check(N.lib().sfx_texture_build_mipmaps(self.handle))

    def read_level(self, level: int) -> np.ndarray:
        """(height >> level, width >> level, components) of a built chain, row 0 = bottom"""
        w, h = max(1, self.size[0] >> level), max(1, self.size[1] >> level)
        out = np.empty((h, w, self.components), self.dtype)
        N.check(N.lib().sfx_texture_read_level(self.handle, level, out.ctypes.data, out.nbytes))
        return out

    def write(self, data, viewport: Optional[tuple[int, int, int, int]] = None) -> None:
        buffer = np.frombuffer(data, np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else np.ascontiguousarray(data).view(np.uint8).ravel()
        x, y, w, h = viewport or (0, 0, 0, 0)
        N.check(N.lib().sfx_texture_write(self.handle, buffer.ctypes.data, buffer.size, x, y, w, h))

    def read(self) -> np.ndarray:
        """(height, width, components), row 0 = bottom"""
        out = np.empty((self.size[1], self.size[0], self.components), self.dtype)
        N.check(N.lib().sfx_texture_read(self.handle, out.ctypes.data, out.nbytes))
        return out

    def device_ptr(self) -> int:
        ptr = C.c_void_p()
        N.check(N.lib().sfx_texture_device_ptr(self.handle, C.byref(ptr), None))
        return ptr.value

    def release(self) -> None:
        if self.handle is not None and self.handle.value:
            N.lib().sfx_texture_destroy(self.handle)
            self.handle = N.Handle()


@define(eq=False, slots=False)
class TextureBox:
    texture: DeviceTexture = None
    data: bytes = field(default=None, repr=False)
    clear: bool = False
    empty: bool = True

    @property
    def fbo(self) -> DeviceTexture:
        """Rendering into a box targets its own texture (the reference pairs every texture with an FBO)"""
        return self.texture

    def release(self) -> None:
        if self.texture is not None:
            self.texture.release()
            self.texture = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass
