"""
ShaderTexture: a `temporal x layers` matrix of device textures that shaders render into and sample from.

Host mirror of the reference's shaderflow/texture.py:74-381 with the moderngl objects replaced by handles of
libshaderflow_hip (sfx_texture_*): same attributes (`final`, `track`, `filter`, `repeat_x/y`, `components`,
`dtype`, `temporal`, `layers`, `width/height/size`), same re-creation rules (`make` on any shape change, cached
bytes re-uploaded when the size is unchanged, :268-270), same write/roll/clear/from_numpy semantics (rows are
flipped so that row 0 is the bottom one, :327-335) and the same uniforms (`<name>Size/Layers/Temporal` and one
sampler per box named `<name>{t}x{l}`, :374-381).
"""
from __future__ import annotations

import ctypes as C
import itertools
from collections import deque
from collections.abc import Iterable
from enum import Enum
from typing import Any, Optional

import numpy as np
from attrs import Factory, define, field

from shaderflow_amd import _native as N
from shaderflow_amd.message import ShaderMessage
from shaderflow_amd.module import ShaderModule
from shaderflow_amd.variable import ShaderVariable, Uniform


class TextureFilter(Enum):
    Nearest = "nearest"
    Linear = "linear"


class Anisotropy(Enum):
    x1 = 1
    x2 = 2
    x4 = 4
    x8 = 8
    x16 = 16


@define(eq=False, slots=False)
class DeviceTexture:
    """What a sampler uniform carries: one native texture (stands for moderngl.Texture)"""
    context: N.Context
    handle: N.Handle
    size: tuple[int, int]
    components: int
    dtype: np.dtype

    @property
    def nbytes(self) -> int:
        return self.size[0]*self.size[1]*self.components*self.dtype.itemsize

    def params(self, filter: str, repeat_x: bool, repeat_y: bool) -> None:
        N.check(N.lib().sfx_texture_params(self.handle, N.LINEAR if filter == "linear" else N.NEAREST, int(repeat_x), int(repeat_y)))

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


def _grow_or_shrink(data, fill, length: int):
    while (len(data) > length):
        data.pop()
    while (len(data) < length):
        data.append(fill())
    return data


@define(eq=False, slots=False)
class ShaderTexture(ShaderModule):
    name: str = None

    def build(self):
        self.make()

    def _changed(self, attr, value, then) -> Any:
        if (converter := attr.converter):
            value = converter(value)
        if getattr(self, attr.name) != value:
            self.__dict__[attr.name] = value
            then()
        return value

    def _apply_on_change(self, attr, value) -> Any:
        return self._changed(attr, value, self.apply)

    def _make_on_change(self, attr, value) -> Any:
        return self._changed(attr, value, self.make)

    final: bool = field(default=False, converter=bool)
    """Bound to the scene's final (resolved) frame: tracks `scene.resolution` instead of the render resolution"""

    track: float = field(default=0.0, converter=float, on_setattr=_make_on_change)
    filter: TextureFilter = field(default=TextureFilter.Linear, converter=TextureFilter, on_setattr=_apply_on_change)
    anisotropy: Anisotropy = field(default=Anisotropy.x16, converter=Anisotropy, on_setattr=_apply_on_change)
    mipmaps: bool = field(default=False, converter=bool, on_setattr=_apply_on_change)
    repeat_x: bool = field(default=True, converter=bool, on_setattr=_apply_on_change)
    repeat_y: bool = field(default=True, converter=bool, on_setattr=_apply_on_change)

    def repeat(self, value: bool):
        self.repeat_x = self.repeat_y = bool(value)
        return self.apply()

    _width: int = field(default=1, converter=int)
    _height: int = field(default=1, converter=int)

    @property
    def width(self) -> int:
        return self.resolution[0] if self.track else self._width

    @width.setter
    def width(self, value: int):
        if (self._width != value):
            self._width = value
            self.make()

    @property
    def height(self) -> int:
        return self.resolution[1] if self.track else self._height

    @height.setter
    def height(self, value: int):
        if (self._height != value):
            self._height = value
            self.make()

    components: int = field(default=4, converter=int, on_setattr=_make_on_change)
    dtype: np.dtype = field(default=np.uint8, converter=np.dtype, on_setattr=_make_on_change)

    @property
    def resolution(self) -> tuple[int, int]:
        if not self.track:
            return (self._width, self._height)
        base = self.scene.resolution if self.final else self.scene.render_resolution      # texture.py:188-192
        return tuple(max(1, int(x*self.track)) for x in base)

    @resolution.setter
    def resolution(self, value: tuple[int, int]):
        if not self.track:
            self.width, self.height = value

    @property
    def size(self) -> tuple[int, int]:
        return self.resolution

    @size.setter
    def size(self, value: tuple[int, int]):
        self.resolution = value

    @property
    def aspect_ratio(self) -> float:
        return self.width/(self.height or 1)

    @property
    def zeros(self) -> np.ndarray:
        return np.zeros((*self.size, self.components), dtype=self.dtype)

    @property
    def bytes_per_pixel(self) -> int:
        return (self.dtype.itemsize*self.components)

    @property
    def size_t(self) -> int:
        return (self.width*self.height*self.bytes_per_pixel)

    matrix: deque = Factory(deque)
    temporal: int = field(default=1, converter=int, on_setattr=_make_on_change)
    layers: int = field(default=1, converter=int, on_setattr=_make_on_change)

    @property
    def boxes(self) -> Iterable[tuple[int, int, TextureBox]]:
        for it, row in enumerate(self.matrix):
            for ib, box in enumerate(row):
                yield (it, ib, box)

    def row(self, n: int = 0) -> Iterable[TextureBox]:
        yield from self.matrix[n]

    def make(self):
        """(Re)allocates every box at the current size/format (texture.py:250-272)"""
        context = self.scene.context
        limit = context.info().max_texture_dim
        if (max(self.size) > limit):
            raise Exception(f"Texture size too large for this context: {self.size} > {limit}")
        if self.dtype not in N.NUMPY_DTYPES:
            raise TypeError(f"Texture dtype {self.dtype} has no device format (uint8, uint16, float32)")

        for row in _grow_or_shrink(self.matrix, deque, self.temporal):
            _grow_or_shrink(row, TextureBox, self.layers)

        for (_, _, box) in self.boxes:
            box.release()
            handle = N.Handle()
            N.check(N.lib().sfx_texture_create(context.handle, self.size[0], self.size[1], self.components,
                                               N.NUMPY_DTYPES[self.dtype], C.byref(handle)))
            box.texture = DeviceTexture(context, handle, self.size, self.components, self.dtype)
            if box.data and (self.size_t == len(box.data)):
                box.texture.write(box.data)
        return self.apply()

    def apply(self):
        for (_, _, box) in self.boxes:
            if box.texture is not None:
                box.texture.params(self.filter.value, self.repeat_x, self.repeat_y)
        return self

    def destroy(self) -> None:
        for (_, _, box) in self.boxes:
            box.release()

    def get_box(self, temporal: int = 0, layer: int = -1) -> Optional[TextureBox]:
        return self.matrix[temporal][layer]

    @property
    def fbo(self) -> DeviceTexture:
        return self.get_box().fbo

    @property
    def texture(self) -> DeviceTexture:
        return self.get_box().texture

    def roll(self, n: int = 1):
        self.matrix.rotate(n)
        return self

    def write(self, data=None, *, temporal: int = 0, layer: int = -1, viewport: tuple[int, int, int, int] = None):
        box = self.get_box(temporal, layer)
        box.texture.write(data, viewport=viewport)
        if (not viewport):
            box.data = bytes(data) if not isinstance(data, np.ndarray) else data.tobytes()
        box.empty = False
        return self

    def from_numpy(self, data: np.ndarray):
        shape = list(data.shape)
        if len(shape) == 2:
            shape.append(1)
        self._height, self._width = shape[0], shape[1]
        self.__dict__["components"] = int(shape[2])
        self.__dict__["dtype"] = np.dtype(data.dtype)
        self.make()
        self.write(np.flipud(data).tobytes())
        return self

    def from_image(self, image):
        from PIL import Image
        return self.from_numpy(np.array(Image.open(image)))

    def clear(self, temporal: int = 0, layer: int = -1):
        return self.write(self.zeros, temporal=temporal, layer=layer)

    def is_empty(self, temporal: int = 0, layer: int = -1) -> bool:
        return self.get_box(temporal, layer).empty

    # module ------------------------------------------------------------------------------------------

    def _coord2name(self, temporal: int, layer: int) -> str:
        return f"{self.name}{temporal}x{layer}"

    def defines(self) -> Iterable[str]:
        """The GLSL the reference would inject (texture.py:351-368); informational here"""
        if not self.name:
            return
        for temporal in range(self.temporal):
            yield f"#define {self.name}{temporal or ''} {self.name}{temporal}x{self.layers-1}"
        yield f"vec4 {self.name}Texture(int temporal, int layer, vec2 astuv) {{"
        for (temporal, layer) in itertools.product(range(self.temporal), range(self.layers)):
            yield f"    if (temporal == {temporal} && layer == {layer})"
            yield f"        return texture({self._coord2name(temporal, layer)}, astuv);"
        yield "    return vec4(0.0);"
        yield "}"

    def handle(self, message):
        if self.track and isinstance(message, ShaderMessage.Shader.RecreateTextures):
            self.make()

    def pipeline(self) -> Iterable[ShaderVariable]:
        if not self.name:
            return
        yield Uniform("vec2", f"{self.name}Size", self.size)
        yield Uniform("int", f"{self.name}Layers", self.layers)
        yield Uniform("int", f"{self.name}Temporal", self.temporal)
        for (it, ib, box) in self.boxes:
            yield Uniform("sampler2D", self._coord2name(it, ib), box.texture)
