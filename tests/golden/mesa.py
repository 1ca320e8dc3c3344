"""
Headless desktop OpenGL (Mesa llvmpipe, OpenGL 4.5 core profile) over ctypes — used ONLY by the golden-vector generators in this
directory (build container; nothing at test time imports this).

`mesa_shim.c` loads `swrast_dri.so` through the DRI software-rasteriser interface (no X server, no EGL); this module compiles it on
first use, resolves GL entry points through `_glapi_get_proc_address` and offers

  * `gl`       — the raw entry points (`gl.glTexImage2D(...)`), resolved lazily;
  * `Context`  — the small program / texture / draw helper the older SwiftShader generator uses (`gles.py`), same methods, but for
                 `#version 330` sources as they are: uniforms are set by their DECLARED type (glGetActiveUniform), a VAO is bound;
  * `refhost.py` builds a stand-in for the un-vendored `moderngl` package on top, so that the reference's own Python drives it.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
SHIM = ROOT/"build"/"mesa"/"libmesa_shim.so"
DRIVER = Path("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so")
GLAPI = Path("/usr/lib/x86_64-linux-gnu/libglapi.so.0")

GL = dict(FRAGMENT_SHADER=0x8B30, VERTEX_SHADER=0x8B31, COMPILE_STATUS=0x8B81, LINK_STATUS=0x8B82, TEXTURE_2D=0x0DE1,
          TEXTURE0=0x84C0, RGBA=0x1908, RGB=0x1907, RG=0x8227, RED=0x1903, UNSIGNED_BYTE=0x1401, UNSIGNED_SHORT=0x1403, FLOAT=0x1406,
          HALF_FLOAT=0x140B, RGBA8=0x8058, RGB8=0x8051, RG8=0x822B, R8=0x8229, RGBA32F=0x8814, RGB32F=0x8815, RG32F=0x8230, R32F=0x822E,
          RGBA16F=0x881A, RGB16F=0x881B, RG16F=0x822F, R16F=0x822D, RGBA16UI=0x8D76, RGB16UI=0x8D77, RG16UI=0x823A, R16UI=0x8234,
          RED_INTEGER=0x8D94, RG_INTEGER=0x8228, RGB_INTEGER=0x8D98, RGBA_INTEGER=0x8D99,
          NEAREST=0x2600, LINEAR=0x2601, NEAREST_MIPMAP_NEAREST=0x2700, LINEAR_MIPMAP_LINEAR=0x2703,
          TEXTURE_MIN_FILTER=0x2801, TEXTURE_MAG_FILTER=0x2800, TEXTURE_WRAP_S=0x2802, TEXTURE_WRAP_T=0x2803,
          TEXTURE_MAX_ANISOTROPY=0x84FE, MAX_TEXTURE_MAX_ANISOTROPY=0x84FF, TEXTURE_BASE_LEVEL=0x813C, TEXTURE_MAX_LEVEL=0x813D,
          REPEAT=0x2901, CLAMP_TO_EDGE=0x812F, FRAMEBUFFER=0x8D40, COLOR_ATTACHMENT0=0x8CE0, FRAMEBUFFER_COMPLETE=0x8CD5,
          TRIANGLE_STRIP=0x0005, ARRAY_BUFFER=0x8892, PIXEL_PACK_BUFFER=0x88EB, STATIC_DRAW=0x88E4, DYNAMIC_DRAW=0x88E8,
          UNPACK_ALIGNMENT=0x0CF5, PACK_ALIGNMENT=0x0D05, COLOR_BUFFER_BIT=0x4000, VERSION=0x1F02, RENDERER=0x1F01,
          SHADING_LANGUAGE_VERSION=0x8B8C, MAX_VIEWPORT_DIMS=0x0D3A, MAX_TEXTURE_SIZE=0x0D33, ACTIVE_UNIFORMS=0x8B86,
          ACTIVE_ATTRIBUTES=0x8B89, SCISSOR_TEST=0x0C11, BLEND=0x0BE2, DEPTH_TEST=0x0B71, CULL_FACE=0x0B44,
          INT=0x1404, BOOL=0x8B56, FLOAT_VEC2=0x8B50, FLOAT_VEC3=0x8B51, FLOAT_VEC4=0x8B52, INT_VEC2=0x8B53, INT_VEC3=0x8B54,
          INT_VEC4=0x8B55, FLOAT_MAT2=0x8B5A, FLOAT_MAT3=0x8B5B, FLOAT_MAT4=0x8B5C, SAMPLER_2D=0x8B5E, UNSIGNED_INT=0x1405,
          READ_ONLY=0x88B8)

_FLOAT_ARGS = {
    "glUniform1f": [C.c_int] + [C.c_float], "glUniform2f": [C.c_int] + [C.c_float]*2, "glUniform3f": [C.c_int] + [C.c_float]*3,
    "glUniform4f": [C.c_int] + [C.c_float]*4, "glClearColor": [C.c_float]*4, "glTexParameterf": [C.c_uint, C.c_uint, C.c_float],
}
_POINTER_ARGS = {
    "glVertexAttribPointer": [C.c_uint, C.c_int, C.c_uint, C.c_ubyte, C.c_int, C.c_void_p],
    "glBufferData": [C.c_uint, C.c_ssize_t, C.c_void_p, C.c_uint],
    "glBufferSubData": [C.c_uint, C.c_ssize_t, C.c_ssize_t, C.c_void_p],
    "glGetBufferSubData": [C.c_uint, C.c_ssize_t, C.c_ssize_t, C.c_void_p],
    "glTexImage2D": [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p],
    "glTexSubImage2D": [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p],
    "glReadPixels": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p],
    "glGetUniformLocation": [C.c_uint, C.c_char_p], "glGetAttribLocation": [C.c_uint, C.c_char_p],
    "glUniformMatrix2fv": [C.c_int, C.c_int, C.c_ubyte, C.c_void_p], "glUniformMatrix3fv": [C.c_int, C.c_int, C.c_ubyte, C.c_void_p],
    "glUniformMatrix4fv": [C.c_int, C.c_int, C.c_ubyte, C.c_void_p],
}


class _EntryPoints:
    """`gl.glXxx` → a ctypes function over the address `_glapi_get_proc_address` returns"""

    def __init__(self):
        SHIM.parent.mkdir(parents=True, exist_ok=True)
        source = HERE/"mesa_shim.c"
        if not SHIM.exists() or SHIM.stat().st_mtime < source.stat().st_mtime:
            subprocess.run(["gcc", "-O2", "-shared", "-fPIC", str(source), "-o", str(SHIM), "-ldl"], check=True)
        self._shim = C.CDLL(str(SHIM))
        self._shim.mesa_error.restype = C.c_char_p
        self._shim.mesa_proc.restype = C.c_void_p
        if self._shim.mesa_init(str(DRIVER).encode(), str(GLAPI).encode(), 3, 3):       # the reference asks for 3.3 core (shader.py:100)
            raise RuntimeError(f"Mesa: {self._shim.mesa_error().decode()}")
        self._pointer = C.CDLL(None)._FuncPtr

    def __getattr__(self, name: str):
        address = self._shim.mesa_proc(name.encode())
        if not address:
            raise AttributeError(name)
        function = self._pointer(address)
        if name in _FLOAT_ARGS:
            function.argtypes = _FLOAT_ARGS[name]
        if name in _POINTER_ARGS:
            function.argtypes = _POINTER_ARGS[name]
        if name == "glGetString":
            function.restype = C.c_char_p
        if name == "glMapBuffer":
            function.restype = C.c_void_p
        setattr(self, name, function)
        return function


_gl: _EntryPoints | None = None


def entry_points() -> _EntryPoints:
    global _gl
    if _gl is None:
        _gl = _EntryPoints()
        _gl.glPixelStorei(GL["UNPACK_ALIGNMENT"], 1)
        _gl.glPixelStorei(GL["PACK_ALIGNMENT"], 1)
        vao = C.c_uint()
        _gl.glGenVertexArrays(1, C.byref(vao))                   # core profile: a VAO must be bound; helpers below make their own
        _gl.glBindVertexArray(vao)
    return _gl


def compile_program(gl, vertex: str, fragment: str, error=RuntimeError) -> int:
    def shader(kind: int, source: str) -> int:
        handle = gl.glCreateShader(kind)
        text = source.encode()
        pointer = C.c_char_p(text)
        gl.glShaderSource(handle, 1, C.byref(pointer), None)
        gl.glCompileShader(handle)
        status = C.c_int()
        gl.glGetShaderiv(handle, GL["COMPILE_STATUS"], C.byref(status))
        if not status.value:
            log = C.create_string_buffer(65536)
            gl.glGetShaderInfoLog(handle, 65536, None, log)
            raise error(f"GLSL compile error ({'vertex' if kind == GL['VERTEX_SHADER'] else 'fragment'}_shader):\n{log.value.decode()}")
        return handle
    program = gl.glCreateProgram()
    gl.glAttachShader(program, shader(GL["VERTEX_SHADER"], vertex))
    gl.glAttachShader(program, shader(GL["FRAGMENT_SHADER"], fragment))
    gl.glLinkProgram(program)
    status = C.c_int()
    gl.glGetProgramiv(program, GL["LINK_STATUS"], C.byref(status))
    if not status.value:
        log = C.create_string_buffer(65536)
        gl.glGetProgramInfoLog(program, 65536, None, log)
        raise error(f"GLSL link error: {log.value.decode()}")
    return program


def active_uniforms(gl, program: int) -> dict[str, tuple[int, int, int]]:
    """name → (location, GL type, array size)"""
    count = C.c_int()
    gl.glGetProgramiv(program, GL["ACTIVE_UNIFORMS"], C.byref(count))
    found = {}
    for index in range(count.value):
        name, length, size, kind = C.create_string_buffer(256), C.c_int(), C.c_int(), C.c_uint()
        gl.glGetActiveUniform(program, index, 256, C.byref(length), C.byref(size), C.byref(kind), name)
        text = name.value.decode()
        found[text] = (gl.glGetUniformLocation(program, name.value), kind.value, size.value)
    return found


def set_uniform(gl, location: int, kind: int, value) -> None:
    """What moderngl's `uniform.value = …` does: the setter follows the uniform's declared type (the program must be in use)"""
    values = list(np.asarray(value).reshape(-1)) if hasattr(value, "__len__") or isinstance(value, np.ndarray) else [value]
    if kind in (GL["INT"], GL["BOOL"], GL["SAMPLER_2D"], GL["UNSIGNED_INT"]):
        gl.glUniform1i(location, int(values[0]))
    elif kind == GL["FLOAT"]:
        gl.glUniform1f(location, float(values[0]))
    elif kind in (GL["FLOAT_VEC2"], GL["FLOAT_VEC3"], GL["FLOAT_VEC4"]):
        n = {GL["FLOAT_VEC2"]: 2, GL["FLOAT_VEC3"]: 3, GL["FLOAT_VEC4"]: 4}[kind]
        assert len(values) == n, f"uniform of {n} floats given {values}"
        getattr(gl, f"glUniform{n}f")(location, *[float(v) for v in values])
    elif kind in (GL["INT_VEC2"], GL["INT_VEC3"], GL["INT_VEC4"]):
        n = {GL["INT_VEC2"]: 2, GL["INT_VEC3"]: 3, GL["INT_VEC4"]: 4}[kind]
        getattr(gl, f"glUniform{n}i")(location, *[int(v) for v in values])
    elif kind in (GL["FLOAT_MAT2"], GL["FLOAT_MAT3"], GL["FLOAT_MAT4"]):
        n = {GL["FLOAT_MAT2"]: 2, GL["FLOAT_MAT3"]: 3, GL["FLOAT_MAT4"]: 4}[kind]
        array = np.ascontiguousarray(values, np.float32)
        assert array.size == n*n
        getattr(gl, f"glUniformMatrix{n}fv")(location, 1, 0, array.ctypes.data)
    else:
        raise TypeError(f"uniform type {kind:#x} not handled")


class Context:
    """program / texture / draw with gles.py's method signatures, on desktop GL"""

    def __init__(self):
        self.gl = entry_points()
        self.version = self.gl.glGetString(GL["VERSION"]).decode()
        self.renderer = self.gl.glGetString(GL["RENDERER"]).decode()

    def program(self, vertex: str, fragment: str) -> int:
        return compile_program(self.gl, vertex, fragment)

    def texture(self, data: np.ndarray, linear: bool, repeat_x: bool, repeat_y: bool) -> int:
        """data (h, w, c) row 0 = bottom; uint8 → unorm, float32 → float texture"""
        gl = self.gl
        data = np.ascontiguousarray(data)
        h, w, c = data.shape
        if data.dtype == np.uint8:
            internal, fmt, kind = {1: ("R8", "RED"), 2: ("RG8", "RG"), 3: ("RGB8", "RGB"), 4: ("RGBA8", "RGBA")}[c] + ("UNSIGNED_BYTE",)
        else:
            internal, fmt, kind = {1: ("R32F", "RED"), 2: ("RG32F", "RG"), 3: ("RGB32F", "RGB"), 4: ("RGBA32F", "RGBA")}[c] + ("FLOAT",)
        handle = C.c_uint()
        gl.glGenTextures(1, C.byref(handle))
        gl.glBindTexture(GL["TEXTURE_2D"], handle)
        gl.glTexImage2D(GL["TEXTURE_2D"], 0, GL[internal], w, h, 0, GL[fmt], GL[kind], data.ctypes.data)
        mode = GL["LINEAR"] if linear else GL["NEAREST"]
        gl.glTexParameteri(GL["TEXTURE_2D"], GL["TEXTURE_MIN_FILTER"], mode)
        gl.glTexParameteri(GL["TEXTURE_2D"], GL["TEXTURE_MAG_FILTER"], mode)
        gl.glTexParameteri(GL["TEXTURE_2D"], GL["TEXTURE_WRAP_S"], GL["REPEAT"] if repeat_x else GL["CLAMP_TO_EDGE"])
        gl.glTexParameteri(GL["TEXTURE_2D"], GL["TEXTURE_WRAP_T"], GL["REPEAT"] if repeat_y else GL["CLAMP_TO_EDGE"])
        assert gl.glGetError() == 0
        return handle.value

    def draw(self, program: int, width: int, height: int, uniforms: dict, textures: dict, attributes: dict, integers: dict = None,
             region: tuple = None) -> np.ndarray:
        """Fullscreen triangle strip into an RGBA8 target; returns (h, w, 4) uint8, row 0 = bottom. `uniforms` (and `integers`,
        kept for gles.py's callers) are set by declared type; region = (x, y, w, h): scissor + read back that rectangle only"""
        gl = self.gl
        target, fbo = C.c_uint(), C.c_uint()
        gl.glGenTextures(1, C.byref(target))
        gl.glBindTexture(GL["TEXTURE_2D"], target)
        gl.glTexImage2D(GL["TEXTURE_2D"], 0, GL["RGBA8"], width, height, 0, GL["RGBA"], GL["UNSIGNED_BYTE"], None)
        gl.glGenFramebuffers(1, C.byref(fbo))
        gl.glBindFramebuffer(GL["FRAMEBUFFER"], fbo)
        gl.glFramebufferTexture2D(GL["FRAMEBUFFER"], GL["COLOR_ATTACHMENT0"], GL["TEXTURE_2D"], target, 0)
        assert gl.glCheckFramebufferStatus(GL["FRAMEBUFFER"]) == GL["FRAMEBUFFER_COMPLETE"]
        gl.glViewport(0, 0, width, height)
        gl.glUseProgram(program)
        declared = active_uniforms(gl, program)
        for name, value in {**uniforms, **(integers or {})}.items():
            if name in declared:
                location, kind, _ = declared[name]
                set_uniform(gl, location, kind, value)
        for unit, (name, handle) in enumerate(textures.items()):
            if name not in declared:
                continue
            gl.glActiveTexture(GL["TEXTURE0"] + unit)
            gl.glBindTexture(GL["TEXTURE_2D"], handle)
            gl.glUniform1i(declared[name][0], unit)
        vao = C.c_uint()
        gl.glGenVertexArrays(1, C.byref(vao))
        gl.glBindVertexArray(vao)
        keep = []
        for name, array in attributes.items():
            location = gl.glGetAttribLocation(program, name.encode())
            if location < 0:
                continue
            array = np.ascontiguousarray(array, np.float32)
            buffer = C.c_uint()
            gl.glGenBuffers(1, C.byref(buffer))
            gl.glBindBuffer(GL["ARRAY_BUFFER"], buffer)
            gl.glBufferData(GL["ARRAY_BUFFER"], array.nbytes, array.ctypes.data, GL["STATIC_DRAW"])
            gl.glEnableVertexAttribArray(location)
            gl.glVertexAttribPointer(location, array.shape[1], GL["FLOAT"], 0, 0, None)
            keep.append(array)
        x, y, rw, rh = region if region is not None else (0, 0, width, height)
        if region is not None:
            gl.glEnable(GL["SCISSOR_TEST"])
            gl.glScissor(x, y, rw, rh)
        gl.glDrawArrays(GL["TRIANGLE_STRIP"], 0, 4)
        if region is not None:
            gl.glDisable(GL["SCISSOR_TEST"])
        out = np.zeros((rh, rw, 4), np.uint8)
        gl.glReadPixels(x, y, rw, rh, GL["RGBA"], GL["UNSIGNED_BYTE"], out.ctypes.data)
        assert gl.glGetError() == 0, "GL error after draw"
        gl.glDeleteFramebuffers(1, C.byref(fbo))
        gl.glDeleteTextures(1, C.byref(target))
        gl.glDeleteVertexArrays(1, C.byref(vao))
        return out


if __name__ == "__main__":
    ctx = Context()
    print(ctx.version, "|", ctx.renderer)
    vs = "#version 330\nin vec2 p; out vec2 uv; void main(){ gl_Position = vec4(p, 0, 1); uv = (p + 1)/2; }"
    fs = "#version 330\nin vec2 uv; out vec4 c; uniform sampler2D t; uniform int k; void main(){ c = texture(t, uv)*k; }"
    prog = ctx.program(vs, fs)
    tex = ctx.texture(np.arange(4*4*3, dtype=np.uint8).reshape(4, 4, 3)*5, True, True, True)
    quad = np.array([[-1, -1], [-1, 1], [1, -1], [1, 1]], np.float32)
    img = ctx.draw(prog, 8, 8, {"k": 1}, {"t": tex}, {"p": quad})
    print(img[:2, :4, :3].tolist())
