"""
ShaderTexture — a `temporal × layers` grid of device textures that shaders render into and sample from.

Same contract as the reference's shaderflow/texture.py:74-381 (attributes `final, track, filter, anisotropy,
mipmaps, repeat_x/y, width/height/size/resolution, components, dtype, temporal, layers, matrix`; methods
`make, apply, repeat, roll, write, from_numpy, from_image, clear, is_empty, get_box, row, boxes, defines, pipeline`),
with libshaderflow_hip textures (device.py) in place of moderngl objects.

Rules that are observable and therefore kept:
  * a texture with `track != 0` follows the scene: `scene.resolution` when `final`, else
    `scene.render_resolution` (= resolution × ssaa), times `track`, at least 1×1;
  * any change of shape/format re-allocates every box (`make`); a box remembers the bytes of its last FULL
    write and gets them back when the new allocation has the same byte size;
  * sampler state (filter, wrap) is applied to every box (`apply`). With `mipmaps=True` (texture.py:131-137, 277-278) `apply`
    ALSO rebuilds the mip chain of every box from its current level 0 and the minification filter becomes LINEAR_MIPMAP_LINEAR
    (NEAREST_MIPMAP_NEAREST for "nearest"); the level of detail comes from the implicit derivatives of the coordinate (OpenGL 3.3
    §3.8.11, glsl.hpp texture_mipmapped). As in the reference, `write` touches level 0 only: the chain is as old as the last
    `apply` — `from_numpy` runs `make()` → `apply()` BEFORE its `write`, so a texture that is only ever filled by `from_numpy` /
    `from_image` samples an EMPTY (zero) chain when minified, until something applies again (`repeat()`, a filter change).
    That is what the reference does on OpenGL (tests/golden/mip.npz holds both cases), so it is what happens here.
    `anisotropy` is accepted and ignored: it is an extension to OpenGL 3.3 core whose filter is implementation-defined
    (tests/golden/filter.npz pins what llvmpipe's does to the goldens' inputs, and that the goldens are rendered without it);
  * `from_numpy` flips rows so that row 0 is the BOTTOM row, like an OpenGL upload of an image;
  * the uniforms are `<name>Size`, `<name>Layers`, `<name>Temporal` and one sampler per box, `<name>{t}x{l}`.
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
from shaderflow_amd.device import DeviceTexture, TextureBox
from shaderflow_amd.message import ShaderMessage
from shaderflow_amd.module import ShaderModule
from shaderflow_amd.variable import ShaderVariable, Uniform


class TextureFilter(Enum):
    Nearest = "nearest"
    Linear = "linear"


class Anisotropy(Enum):
    x1, x2, x4, x8, x16 = 1, 2, 4, 8, 16


def _on_change(action: str):
    """attrs on_setattr hook: convert, store, and run `self.<action>()` only when the value really changed"""
    def hook(self, attribute, value) -> Any:
        if attribute.converter is not None:
            value = attribute.converter(value)
        if getattr(self, attribute.name) != value:
            self.__dict__[attribute.name] = value
            getattr(self, action)()
        return value
    return hook


@define(eq=False, slots=False)
class ShaderTexture(ShaderModule):
    name: str = None

    # what the texture follows and how it is sampled
    final: bool = field(default=False, converter=bool)
    track: float = field(default=0.0, converter=float, on_setattr=_on_change("make"))
    filter: TextureFilter = field(default=TextureFilter.Linear, converter=TextureFilter, on_setattr=_on_change("apply"))
    anisotropy: Anisotropy = field(default=Anisotropy.x16, converter=Anisotropy, on_setattr=_on_change("apply"))
    mipmaps: bool = field(default=False, converter=bool, on_setattr=_on_change("apply"))
    repeat_x: bool = field(default=True, converter=bool, on_setattr=_on_change("apply"))
    repeat_y: bool = field(default=True, converter=bool, on_setattr=_on_change("apply"))

    # storage
    _width: int = field(default=1, converter=int)
    _height: int = field(default=1, converter=int)
    components: int = field(default=4, converter=int, on_setattr=_on_change("make"))
    dtype: np.dtype = field(default=np.uint8, converter=np.dtype, on_setattr=_on_change("make"))
    temporal: int = field(default=1, converter=int, on_setattr=_on_change("make"))
    layers: int = field(default=1, converter=int, on_setattr=_on_change("make"))
    matrix: deque = Factory(deque)
    """matrix[t][l]: t frames back in time, layer l; rotated by `roll()` after each render"""

    def build(self):
        self.make()

    # size -------------------------------------------------------------------------------------------------

    @property
    def resolution(self) -> tuple[int, int]:
        if not self.track:
            return (self._width, self._height)
        followed = self.scene.resolution if self.final else self.scene.render_resolution
        return tuple(max(1, int(extent*self.track)) for extent in followed)

    @resolution.setter
    def resolution(self, value: tuple[int, int]):
        if not self.track:
            self.width, self.height = value

    size = resolution

    @property
    def width(self) -> int:
        return self.resolution[0]

    @width.setter
    def width(self, value: int):
        if value != self._width:
            self._width = value
            self.make()

    @property
    def height(self) -> int:
        return self.resolution[1]

    @height.setter
    def height(self, value: int):
        if value != self._height:
            self._height = value
            self.make()

    @property
    def aspect_ratio(self) -> float:
        return self.width/(self.height or 1)

    @property
    def bytes_per_pixel(self) -> int:
        return self.components*self.dtype.itemsize

    @property
    def size_t(self) -> int:
        width, height = self.resolution
        return width*height*self.bytes_per_pixel

    @property
    def zeros(self) -> np.ndarray:
        return np.zeros((*self.resolution, self.components), dtype=self.dtype)

    # boxes ------------------------------------------------------------------------------------------------

    @property
    def boxes(self) -> Iterable[tuple[int, int, TextureBox]]:
        for t, row in enumerate(self.matrix):
            for l, box in enumerate(row):
                yield (t, l, box)

    def row(self, n: int = 0) -> Iterable[TextureBox]:
        yield from self.matrix[n]

    def get_box(self, temporal: int = 0, layer: int = -1) -> Optional[TextureBox]:
        return self.matrix[temporal][layer]

    @property
    def texture(self) -> DeviceTexture:
        """Most recent frame, last layer"""
        return self.get_box().texture

    @property
    def fbo(self) -> DeviceTexture:
        return self.get_box().fbo

    def roll(self, n: int = 1):
        self.matrix.rotate(n)
        return self

    def _reshape_matrix(self) -> None:
        while len(self.matrix) > self.temporal:
            for box in self.matrix.pop():
                box.release()
        while len(self.matrix) < self.temporal:
            self.matrix.append(deque())
        for row in self.matrix:
            while len(row) > self.layers:
                row.pop().release()
            while len(row) < self.layers:
                row.append(TextureBox())

    def make(self):
        """(Re)allocate every box at the current size and format"""
        context = self.scene.context
        width, height = self.resolution
        limit = context.info().max_texture_dim
        if max(width, height) > limit:
            raise Exception(f"Texture size too large for this context: {(width, height)} > {limit}")
        if self.dtype not in N.NUMPY_DTYPES:
            raise TypeError(f"Texture dtype {self.dtype} has no device format (uint8, uint16, float16, float32)")
        self._reshape_matrix()
        for (_, _, box) in self.boxes:
            box.release()
            handle = N.Handle()
            N.check(N.lib().sfx_texture_create(context.handle, width, height, self.components,
                                               N.NUMPY_DTYPES[self.dtype], C.byref(handle)))
            box.texture = DeviceTexture(context, handle, (width, height), self.components, self.dtype)
            if box.data and len(box.data) == self.size_t:
                box.texture.write(box.data)
        return self.apply()

    def apply(self):
        """Push filter and wrap state to every box"""
        for (_, _, box) in self.boxes:
            if box.texture is not None:
                if self.mipmaps:
                    box.texture.build_mipmaps()                     # texture.py:277-278: from whatever level 0 holds NOW
                box.texture.params(self.filter.value, self.repeat_x, self.repeat_y, mipmaps=self.mipmaps)
        return self

    def repeat(self, value: bool):
        self.repeat_x = self.repeat_y = bool(value)
        return self.apply()

    def destroy(self) -> None:
        for (_, _, box) in self.boxes:
            box.release()

    # data -------------------------------------------------------------------------------------------------

    def write(self, data=None, *, temporal: int = 0, layer: int = -1, viewport: Optional[tuple[int, int, int, int]] = None):
        box = self.get_box(temporal, layer)
        box.texture.write(data, viewport=viewport)
        if viewport is None:
            box.data = data.tobytes() if isinstance(data, np.ndarray) else bytes(data)
        box.empty = False
        return self

    def clear(self, temporal: int = 0, layer: int = -1):
        return self.write(self.zeros, temporal=temporal, layer=layer)

    def is_empty(self, temporal: int = 0, layer: int = -1) -> bool:
        return self.get_box(temporal, layer).empty

    def from_numpy(self, data: np.ndarray):
        """(height, width[, components]) array, top row first → texture of that shape and dtype"""
        if data.ndim == 2:
            data = data[:, :, None]
        self._height, self._width = int(data.shape[0]), int(data.shape[1])
        self.__dict__["components"] = int(data.shape[2])
        self.__dict__["dtype"] = np.dtype(data.dtype)
        self.make()
        return self.write(np.flipud(data).tobytes())

    def from_image(self, image):
        from PIL import Image
        return self.from_numpy(np.array(Image.open(image)))

    # module -----------------------------------------------------------------------------------------------

    def _sampler_name(self, temporal: int, layer: int) -> str:
        return f"{self.name}{temporal}x{layer}"

    def defines(self) -> Iterable[str]:
        """The GLSL helper text the reference injects for this texture (texture.py:349-363); goes in front of fragments that are translated"""
        if not self.name:
            return
        last = self.layers - 1
        for t in range(self.temporal):
            yield f"#define {self.name}{t or ''} {self.name}{t}x{last}"
        yield f"vec4 {self.name}Texture(int temporal, int layer, vec2 astuv) {{"
        for (t, l) in itertools.product(range(self.temporal), range(self.layers)):
            yield f"    if (temporal == {t} && layer == {l})"
            yield f"        return texture({self._sampler_name(t, l)}, astuv);"
        yield "    return vec4(0.0);"
        yield "}"

    def handle(self, message):
        if self.track and isinstance(message, ShaderMessage.Shader.RecreateTextures):
            self.make()

    def pipeline_token(self):
        # Size / Layers / Temporal and one sampler per box: which DEVICE texture sits in which box, in matrix order (roll() rotates it)
        if type(self).pipeline is not ShaderTexture.pipeline:
            return None                                              # a subclass yields variables of its own: walk it every frame, as the reference does
        return (self.name, self.resolution, tuple(0 if box.texture is None else box.texture.serial for (_, _, box) in self.boxes))

    def pipeline(self) -> Iterable[ShaderVariable]:
        if not self.name:
            return
        yield Uniform("vec2", f"{self.name}Size", self.resolution)
        yield Uniform("int", f"{self.name}Layers", self.layers)
        yield Uniform("int", f"{self.name}Temporal", self.temporal)
        for (t, l, box) in self.boxes:
            yield Uniform("sampler2D", self._sampler_name(t, l), box.texture)
