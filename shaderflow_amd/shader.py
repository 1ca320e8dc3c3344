"""
ShaderProgram: a fragment program that renders a fullscreen quad into its own ShaderTexture.

Host mirror of the reference's shaderflow/shader.py:99-425. What moderngl/OpenGL did there is done here by
libshaderflow_hip: `compile()` resolves the fragment source through the registry of fragments restated as HIP
kernels (`sfx_program_lookup`); a source that is not there is translated to HIP C++, compiled and loaded
(`glsl2hip.py`, `sfx_program_load`), and one that fails to translate or compile gets the `missing` kernel — the
reference's compile-error fallback, shader.py:323-340. `use_pipeline()` pushes every Uniform by name and binds
samplers, `render()` draws each layer (`sfx_render`) and rolls the temporal matrix. The scene's `iFinal` program (texture.final) is the SSAA
resolve (shader.py:391-396): `sfx_resolve`, or fused with the main pass (`sfx_render_resolve`) when
final.glsl's footprint stays inside the output pixel's own supersample block.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import re
import struct
from collections.abc import Iterable
from pathlib import Path
from typing import Any, Optional, Union

import numpy as np
from attrs import Factory, define

import shaderflow_amd
from shaderflow_amd import _native as N
from shaderflow_amd.message import ShaderMessage
from shaderflow_amd.module import ShaderModule
from shaderflow_amd.texture import DeviceTexture, ShaderTexture
from shaderflow_amd.variable import FlatVariable, InVariable, OutVariable, ShaderVariable

_PLAIN_SCALARS = (float, int, bool)
_VECTOR_FAST = True
_UNIFORM_TYPES = {"float": (N.T_FLOAT, np.float32, 1), "int": (N.T_INT, np.int32, 1), "bool": (N.T_BOOL, np.int32, 1),
                  "vec2": (N.T_VEC2, np.float32, 2), "vec3": (N.T_VEC3, np.float32, 3), "vec4": (N.T_VEC4, np.float32, 4),
                  "mat2": (N.T_MAT2, np.float32, 4), "mat3": (N.T_MAT3, np.float32, 9), "mat4": (N.T_MAT4, np.float32, 16)}


@define
class ShaderDumper:
    """What the reference does with a fragment its driver rejects (shader.py:36-97): the texts and the error go to the
    user's log directory, the first error is shown with its surrounding lines. Here the compiler is hipcc and there are
    two texts: the assembled GLSL and its translation (line numbers in the error are the translation's)."""
    shader: "ShaderProgram"
    error: str
    fragment: str
    translation: str = ""
    context: int = 5

    _parser = re.compile(r"^[^\n:]*\.hip:(\d+):(\d+): error: (.*)$", re.MULTILINE)

    @staticmethod
    def directory() -> Path:
        from shaderflow_amd import glsl2hip
        return Path(os.environ.get("SHADERFLOW_LOG_DIR", glsl2hip.cache_directory().parent/"logs"))

    def excerpt(self) -> str:
        """the lines of the translation around the first compiler error, the faulty one marked"""
        match = self._parser.search(self.error)
        if match is None or not self.translation:
            return ""
        lines, lineno = self.translation.splitlines(), int(match.group(1))
        first, last = max(0, lineno - self.context - 1), min(len(lines), lineno + self.context)
        body = [f"({n + 1:3d}) {'>' if n + 1 == lineno else '|'} {lines[n]}" for n in range(first, last)]
        return "\n".join([f"Module #{self.shader.uuid}, line {lineno}: {match.group(3)}", *body])

    def dump(self) -> Path:
        directory = self.directory()
        directory.mkdir(parents=True, exist_ok=True)
        self.shader.log_error(f"Dumping shaders to {directory}")
        (directory/f"{self.shader.uuid}.frag").write_text(self.fragment, encoding="utf-8")
        if self.translation:
            (directory/f"{self.shader.uuid}.hip").write_text(self.translation, encoding="utf-8")
        (directory/f"{self.shader.uuid}-error.md").write_text(self.error, encoding="utf-8")
        excerpt = self.excerpt()
        if excerpt:
            self.shader.log_error(excerpt)
        return directory


@define(eq=False, slots=False)
class ShaderProgram(ShaderModule):
    version: int = 330
    clear: bool = True
    instances: int = 1
    texture: ShaderTexture = None

    def build(self):
        self.texture = ShaderTexture(scene=self.scene, name=self.name, track=True)
        # The varyings every fragment can read (shader.py:112-124); kept as declarations for introspection
        self.fragment_variable(OutVariable("vec4", "fragColor"))
        self.vertex_variable(InVariable("vec2", "vertex_position"))
        self.vertex_variable(InVariable("vec2", "vertex_gluv"))
        for name in ("fragCoord", "stxy", "glxy", "stuv", "astuv", "gluv", "agluv"):
            self.traverse_variable(ShaderVariable("vec2", name))
        self.traverse_variable(FlatVariable("int", "instance"))
        for x in (-1, 1):
            for y in (-1, 1):
                self.add_vertice(x=x, y=y, u=x, v=y)
        self.vertex = (shaderflow_amd.resources/"shaders"/"vertex"/"default.glsl")
        self.fragment = (shaderflow_amd.resources/"shaders"/"fragment"/"default.glsl")

    # variable bookkeeping (shader.py:134-153) ---------------------------------------------------------

    vertex_variables: dict = Factory(dict)
    fragment_variables: dict = Factory(dict)

    def vertex_variable(self, variable: ShaderVariable) -> None:
        self.vertex_variables.setdefault(variable.name, variable)

    def fragment_variable(self, variable: ShaderVariable) -> None:
        self.fragment_variables.setdefault(variable.name, variable)

    def common_variable(self, variable: ShaderVariable) -> None:
        self.fragment_variable(variable)
        self.vertex_variable(variable)

    def traverse_variable(self, variable: ShaderVariable) -> None:
        self.fragment_variable(variable.copy(direction="in"))
        self.vertex_variable(variable.copy(direction="out"))

    vertices: list = Factory(list)

    def add_vertice(self, x: float = 0, y: float = 0, u: float = 0, v: float = 0) -> None:
        self.vertices.extend((x, y, u, v))

    # sources ------------------------------------------------------------------------------------------

    _vertex: Union[Path, str] = ""
    _fragment: Union[Path, str] = ""

    @staticmethod
    def _read(content: Union[Path, str]) -> str:
        if isinstance(content, Path):
            return content.read_text()
        text = str(content)
        # A short string may be a path given as str (shader.py:258-266 tolerates both)
        if ("\n" not in text) and (len(text) < 4096) and text and os.path.exists(text):
            return Path(text).read_text()
        return text

    @property
    def vertex(self) -> str:
        return self._read(self._vertex)

    @vertex.setter
    def vertex(self, value: Union[Path, str]):
        self._vertex = value

    @property
    def fragment(self) -> str:
        """User content of the fragment shader (what the registry fingerprints)"""
        return self._read(self._fragment)

    @fragment.setter
    def fragment(self, value: Union[Path, str]):
        self._fragment = value
        if self.program is not None:                         # hot swap: resolve again on next compile message
            self.release_program()

    # device program -----------------------------------------------------------------------------------

    program: Optional[N.Handle] = None
    _pushed: dict = Factory(dict)
    """name → (bytes, known) of the value the device program holds: unchanged uniforms and samplers are not sent again"""
    _pushed_plain: dict = Factory(dict)
    """name → (type, python scalar, known): the same answer for python scalars before any conversion (`_push`)"""
    _module_tokens: dict = Factory(dict)
    """id(module) → the pipeline_token() this program last consumed that module's pipeline under (use_scene_pipeline)"""
    _module_names: dict = Factory(dict)
    _shared_names: frozenset = frozenset()
    fallback: bool = False
    """True when the fragment was unknown and the `missing` kernel was bound (shader.py:336-340)"""

    def release_program(self) -> None:
        if self.program is not None and self.program.value:
            N.lib().sfx_program_destroy(self.program)
        self.program = None
        self._pushed.clear()
        self._pushed_plain.clear()
        self._module_tokens.clear()
        self._module_names.clear()
        self._shared_names = frozenset()
        self._uniform_arrays = {}

    def compile(self, _vertex: str = None, _fragment: str = None):
        for variable in self.full_pipeline():
            self.common_variable(variable)
        self.release_program()
        source = _fragment or self.fragment
        handle, fallback = N.Handle(), C.c_int(0)
        N.check(N.lib().sfx_program_lookup(self.scene.context.handle, source.encode("utf-8"), C.byref(handle), C.byref(fallback)))
        self.program, self.fallback, self.translated = handle, bool(fallback.value), False
        if self.fallback and self.translate:
            try:
                self._load_translated(source)
            except Exception as error:                       # shader.py:326-340: report, keep rendering with missing.glsl
                from shaderflow_amd.glsl2hip import CompileError, TranslationError
                if not isinstance(error, (TranslationError, CompileError, N.NativeError)):
                    raise
                self.compile_error = str(error)
                self.log_error(f"Fragment could not be translated: {str(error).splitlines()[0]}")
                ShaderDumper(shader=self, error=str(error), fragment=self.assembled_fragment(source), translation=self._translation_text).dump()
        if self.fallback:
            self.log_error("Fragment is neither in the kernel registry nor translatable, loading missing texture shader")
        return self

    translate: bool = os.environ.get("SHADERFLOW_TRANSLATE", "1") != "0"
    """Fragments that are not in the registry are translated to HIP and compiled at run time (glsl2hip.py)"""
    translated: bool = False
    compile_error: str = ""
    _translation_text: str = ""
    _uniform_arrays: dict = Factory(dict)

    def assembled_fragment(self, content: str) -> str:
        """What the reference hands to the GLSL compiler after the declarations and the prelude (shader.py:214-233): every
        module's defines and includes, then the content. A module's helper functions are only included when the content
        names them (they would otherwise make every box of every texture an active sampler)."""
        parts: list[str] = []
        for module in self.scene.modules:
            lines = [line for line in module.defines() if line]
            parts.extend(line for line in lines if line.lstrip().startswith("#"))
            helpers = [line for line in lines if not line.lstrip().startswith("#")]
            if helpers:
                names = re.findall(r"\b(\w+)\s*\(", helpers[0])
                if any(re.search(rf"\b{re.escape(name)}\b", content) for name in names):
                    parts.extend(helpers)
            for include in filter(None, module.includes()):
                parts.append(include.read_text() if isinstance(include, Path) else str(include))
        parts.append(content)
        return "\n".join(parts)

    def _load_translated(self, content: str) -> None:
        from shaderflow_amd import glsl2hip
        variables = [(variable.type, variable.name) for variable in self.full_pipeline()]
        self._translation_text = ""
        translation = glsl2hip.translate(self.assembled_fragment(content), variables)
        self._translation_text = translation.cpp
        code = glsl2hip.compile(translation)
        keep = [binding.name.encode() for binding in translation.bindings]
        table = (N.Binding*max(1, len(keep)))(*[N.Binding(name, int(binding.sampler), binding.slot, binding.count, int(binding.integer))
                                                for name, binding in zip(keep, translation.bindings)])
        handle = N.Handle()
        N.check(N.lib().sfx_program_load(self.scene.context.handle, code, len(code), table, len(keep), C.byref(handle)))
        self.release_program()
        self.program, self.fallback, self.translated = handle, False, True
        # uniform arrays are set through their elements (`name[i]`, as GL names them); initialisers of `uniform T x = …;`
        # declarations are what the program reads until the host sets the uniform (GLSL 3.30 §4.3.5)
        self._uniform_arrays = {}
        for binding in translation.bindings:
            if binding.array:
                kind, length = self._uniform_arrays.get(binding.array, (binding.type, 0))
                self._uniform_arrays[binding.array] = (kind, length + 1)
            if binding.default is not None:
                self._push(binding.name, binding.default, binding.type)

    @property
    def kernel(self) -> str:
        """Registry name of the bound kernel"""
        return N.lib().sfx_program_name(self.program).decode() if self.program is not None else ""

    def set_uniform(self, name: str, value: Any = None) -> None:
        if (self.program is None):
            raise RuntimeError("Shader hasn't been compiled yet")
        if (value is None):
            return
        for key, yielded in self._module_names.items():              # a module that yields this name says its own value again next frame
            if name in yielded:
                self._module_tokens.pop(key, None)
        self._push(name, value)

    def _push(self, name: str, value: Any, type: Optional[str] = None) -> bool:
        # Most uniforms of most frames are python scalars that did not change: answered from the last (type, value) pair without a
        # numpy round trip. 0.0 and -0.0 compare equal but are different bits, and NaN never equals itself: both take the long way.
        if type is not None and value.__class__ in _PLAIN_SCALARS:
            last = self._pushed_plain.get(name)
            if (last is not None and last[0] is type and last[1].__class__ is value.__class__ and last[1] == value and
                    (value != 0 or value.__class__ is not float or math.copysign(1.0, value) == math.copysign(1.0, last[1]))):
                return last[2]
            known = self._push_converted(name, value, type)
            self._pushed_plain[name] = (type, value, known)
            return known
        if type is not None and _VECTOR_FAST:
            # vectors: numpy arrays by their bytes, tuples and lists of python numbers by their float64 image
            key = None
            if value.__class__ is np.ndarray:
                key = (value.dtype.str, value.tobytes())
            elif value.__class__ in (tuple, list) and len(value) <= 16:
                try:
                    key = struct.pack(f"{len(value)}d", *value)
                except (struct.error, TypeError):
                    key = None
            if key is not None:
                last = self._pushed_plain.get(name)
                if last is not None and last[0] is type and last[1] == key:
                    return last[2]
                known = self._push_converted(name, value, type)
                self._pushed_plain[name] = (type, key, known)
                return known
        return self._push_converted(name, value, type)

    def _push_converted(self, name: str, value: Any, type: Optional[str] = None) -> bool:
        if name in self._uniform_arrays:                              # `uniform T name[N]` of a translated fragment: element by element
            kind, length = self._uniform_arrays[name]
            elements = np.asarray(value, dtype=np.float64).reshape(length, -1)
            return all([self._push_converted(f"{name}[{index}]", elements[index], kind) for index in range(length)])
        if type is None:
            array = np.asarray(value)
            count = array.size
            if array.dtype.kind in "biu":
                type = "int" if count == 1 else {2: "vec2", 3: "vec3", 4: "vec4"}[count]
            else:
                type = {1: "float", 2: "vec2", 3: "vec3", 4: "vec4"}.get(count)
            if type is None:
                return False
        code, dtype, count = _UNIFORM_TYPES[type]
        data = np.ascontiguousarray(np.asarray(value, dtype=np.float64).ravel()[:count].astype(dtype))
        if data.size < count:
            return False
        raw = (code, data.tobytes())
        previous = self._pushed.get(name)
        if previous is not None and previous[0] == raw:
            return previous[1]
        known = C.c_int(0)
        N.check(N.lib().sfx_uniform_set(self.program, name.encode(), code, data.ctypes.data, C.byref(known)))
        self._pushed[name] = (raw, bool(known.value))
        self._pushed_plain.pop(name, None)                            # whoever asked through `_push` records its own value again
        return bool(known.value)

    def get_uniform(self, name: str) -> Optional[Any]:
        return None

    SKIP_GPU: bool = os.environ.get("SKIP_GPU") == "1"

    def use_pipeline(self, pipeline: Iterable[ShaderVariable], *, _index: int = 0) -> None:
        for variable in pipeline:
            if (variable.type == "sampler2D"):
                texture = variable.value
                if isinstance(texture, DeviceTexture) and texture.handle.value:
                    bound = ("sampler", texture.serial)
                    if self._pushed.get(variable.name) != bound:
                        N.check(N.lib().sfx_sampler_bind(self.program, variable.name.encode(), texture.handle, None))
                        self._pushed[variable.name] = bound
                _index += 1
                continue
            if variable.value is None or variable.type not in _UNIFORM_TYPES:
                continue
            self._push(variable.name, variable.value, variable.type)

    def use_scene_pipeline(self) -> None:
        """`use_pipeline(full_pipeline())` (shader.py:377-385) without walking modules whose variables cannot have changed: a module
        that answers `pipeline_token()` with the value this program saw when it last consumed the module's pipeline is skipped —
        unless a module walked earlier in this frame yields a name the skipped one yields too (the later module's value must win,
        as it does when everything is walked in order). Modules without a token are walked every frame."""
        tokens, names = self._module_tokens, self._module_names
        touched: set = set()                                          # shared names pushed by modules walked so far this frame
        shared = self._shared_names
        for module in self.scene.modules:
            key = id(module)
            token = module.pipeline_token()
            if token is not None and key in tokens and tokens[key] == token and not (shared and not touched.isdisjoint(names[key])):
                continue
            variables = list(module.pipeline() or ())
            self.use_pipeline(variables)
            tokens[key] = token
            yielded = frozenset(variable.name for variable in variables)
            if names.get(key) != yielded:
                names[key] = yielded
                seen: set = set()
                again: set = set()
                for other in names.values():
                    again |= (seen & other)
                    seen |= other
                self._shared_names = shared = frozenset(again)
            if shared:
                touched |= (yielded & shared)

    def render_to_fbo(self, fbo: DeviceTexture, clear: bool = True, layer: int = 0) -> None:
        if self.SKIP_GPU:
            return
        N.check(N.lib().sfx_render(self.program, fbo.handle, layer))

    def render(self, pipeline: bool = True) -> None:
        """`pipeline=False`: the caller has set the uniforms and samplers this frame changes (clockloop.py)"""
        if self.program is None:
            self.compile()
        # host-only frames of a sharded export (parallel.frame_modes mode 0): the temporal matrix still rolls
        skip = self.SKIP_GPU or self.scene._skip_render

        if self.texture.final:
            # shader.py:391-396 — iScreen (RGBA8, linear, clamp) → iFinal (RGB8)
            if skip or self.scene._fused_this_frame:
                return None
            source = self.scene.shader.texture.texture
            N.check(N.lib().sfx_resolve(self.scene.context.handle, source.handle, self.texture.fbo.handle, self.scene.subsample))
            return None

        if not skip and pipeline:
            self.use_scene_pipeline()

        # Main pass + resolve in one kernel when final.glsl only needs the pixel's own supersamples
        if (self is self.scene.shader) and self.scene._can_fuse(self):
            if not skip:
                N.check(N.lib().sfx_render_resolve(self.program, self.scene._final.texture.fbo.handle,
                                                   int(self.scene.ssaa), self.scene.subsample))
            self.scene._fused_this_frame = True
            self.texture.roll()
            return None

        if not skip:
            for layer, box in enumerate(self.texture.row(0)):
                self.set_uniform("iLayer", layer)
                self.render_to_fbo(fbo=box.fbo, clear=box.clear, layer=layer)
        self.texture.roll()

    def update(self) -> None:
        self.render()

    def handle(self, message) -> None:
        if isinstance(message, ShaderMessage.Shader.Compile):
            self.compile()
        elif isinstance(message, ShaderMessage.Shader.Render):
            self.render()

    def destroy(self) -> None:
        self.release_program()
