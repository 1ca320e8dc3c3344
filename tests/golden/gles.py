"""
A minimal headless OpenGL ES 3.0 harness over ctypes (EGL surfaceless context + FBO), used ONLY by the golden-vector
generators in this directory. The implementation is whatever `libEGL.so`/`libGLESv2.so` the build container offers —
here Google SwiftShader 4.1 (a CPU rasteriser shipped inside the `kaleido` wheel); nothing at test time imports this.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

SWIFTSHADER = Path("/usr/local/lib/python3.10/dist-packages/kaleido/executable/bin/swiftshader")

EGL_NONE, EGL_OPENGL_ES_API, EGL_CONTEXT_CLIENT_VERSION = 0x3038, 0x30A0, 0x3098
EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT = 0x3033, 0x0001, 0x3040, 0x0040
EGL_WIDTH, EGL_HEIGHT = 0x3057, 0x3056
GL = dict(FRAGMENT_SHADER=0x8B30, VERTEX_SHADER=0x8B31, COMPILE_STATUS=0x8B81, LINK_STATUS=0x8B82, TEXTURE_2D=0x0DE1,
          TEXTURE0=0x84C0, RGBA=0x1908, RGB=0x1907, RG=0x8227, RED=0x1903, UNSIGNED_BYTE=0x1401, FLOAT=0x1406,
          RGBA8=0x8058, RGB8=0x8051, RG8=0x822B, R8=0x8229, RGBA32F=0x8814, RG32F=0x8230, R32F=0x822E,
          NEAREST=0x2600, LINEAR=0x2601, TEXTURE_MIN_FILTER=0x2801, TEXTURE_MAG_FILTER=0x2800, TEXTURE_WRAP_S=0x2802, TEXTURE_WRAP_T=0x2803,
          REPEAT=0x2901, CLAMP_TO_EDGE=0x812F, FRAMEBUFFER=0x8D40, COLOR_ATTACHMENT0=0x8CE0, FRAMEBUFFER_COMPLETE=0x8CD5,
          TRIANGLE_STRIP=0x0005, ARRAY_BUFFER=0x8892, STATIC_DRAW=0x88E4, UNPACK_ALIGNMENT=0x0CF5, PACK_ALIGNMENT=0x0D05,
          COLOR_BUFFER_BIT=0x4000, VERSION=0x1F02, RENDERER=0x1F01, EXTENSIONS=0x1F03)


class Context:
    def __init__(self):
        self.gl = C.CDLL(str(SWIFTSHADER/"libGLESv2.so"), mode=C.RTLD_GLOBAL)
        self.egl = C.CDLL(str(SWIFTSHADER/"libEGL.so"), mode=C.RTLD_GLOBAL)
        egl = self.egl
        egl.eglGetDisplay.restype = C.c_void_p; egl.eglGetDisplay.argtypes = [C.c_void_p]
        egl.eglInitialize.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        egl.eglChooseConfig.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
        egl.eglCreateContext.restype = C.c_void_p; egl.eglCreateContext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        egl.eglCreatePbufferSurface.restype = C.c_void_p; egl.eglCreatePbufferSurface.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        egl.eglMakeCurrent.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.display = egl.eglGetDisplay(None)
        major, minor = C.c_int(), C.c_int()
        assert egl.eglInitialize(self.display, C.byref(major), C.byref(minor))
        egl.eglBindAPI(EGL_OPENGL_ES_API)
        attribs = (C.c_int*5)(EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT, EGL_NONE)[:]
        config, count = C.c_void_p(), C.c_int()
        assert egl.eglChooseConfig(self.display, (C.c_int*5)(*attribs), C.byref(config), 1, C.byref(count)) and count.value
        self.context = egl.eglCreateContext(self.display, config, None, (C.c_int*3)(EGL_CONTEXT_CLIENT_VERSION, 3, EGL_NONE))
        assert self.context
        surface = egl.eglCreatePbufferSurface(self.display, config, (C.c_int*5)(EGL_WIDTH, 16, EGL_HEIGHT, 16, EGL_NONE))
        assert egl.eglMakeCurrent(self.display, surface, surface, self.context)
        gl = self.gl
        gl.glGetString.restype = C.c_char_p
        gl.glGetUniformLocation.argtypes = [C.c_uint, C.c_char_p]
        gl.glGetAttribLocation.argtypes = [C.c_uint, C.c_char_p]
        for name in ("glUniform1f", "glUniform2f", "glUniform3f", "glUniform4f"):
            getattr(gl, name).argtypes = [C.c_int] + [C.c_float]*int(name[9])
        gl.glVertexAttribPointer.argtypes = [C.c_uint, C.c_int, C.c_uint, C.c_ubyte, C.c_int, C.c_void_p]
        gl.glBufferData.argtypes = [C.c_uint, C.c_ssize_t, C.c_void_p, C.c_uint]
        gl.glTexImage2D.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]
        gl.glReadPixels.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]
        gl.glPixelStorei(GL["UNPACK_ALIGNMENT"], 1); gl.glPixelStorei(GL["PACK_ALIGNMENT"], 1)
        self.version = gl.glGetString(GL["VERSION"]).decode()
        self.renderer = gl.glGetString(GL["RENDERER"]).decode()
        self.extensions = (gl.glGetString(GL["EXTENSIONS"]) or b"").decode().split()

    # shaders ----------------------------------------------------------------------------------------------
    def _shader(self, kind: int, source: str) -> int:
        gl = self.gl
        shader = gl.glCreateShader(kind)
        text = source.encode()
        pointer = C.c_char_p(text)
        gl.glShaderSource(shader, 1, C.byref(pointer), None)
        gl.glCompileShader(shader)
        status = C.c_int()
        gl.glGetShaderiv(shader, GL["COMPILE_STATUS"], C.byref(status))
        if not status.value:
            log = C.create_string_buffer(16384)
            gl.glGetShaderInfoLog(shader, 16384, None, log)
            numbered = "\n".join(f"{n + 1:4d} {line}" for n, line in enumerate(source.splitlines()))
            raise RuntimeError(f"GLSL ES compile error:\n{log.value.decode()}\n{numbered}")
        return shader

    def program(self, vertex: str, fragment: str) -> int:
        gl = self.gl
        program = gl.glCreateProgram()
        gl.glAttachShader(program, self._shader(GL["VERTEX_SHADER"], vertex))
        gl.glAttachShader(program, self._shader(GL["FRAGMENT_SHADER"], fragment))
        gl.glLinkProgram(program)
        status = C.c_int()
        gl.glGetProgramiv(program, GL["LINK_STATUS"], C.byref(status))
        if not status.value:
            log = C.create_string_buffer(16384)
            gl.glGetProgramInfoLog(program, 16384, None, log)
            raise RuntimeError(f"GLSL ES link error: {log.value.decode()}")
        return program

    # textures ---------------------------------------------------------------------------------------------
    def texture(self, data: np.ndarray, linear: bool, repeat_x: bool, repeat_y: bool) -> int:
        """data (h, w, c) row 0 = bottom; uint8 → unorm, float32 → float texture"""
        gl = self.gl
        data = np.ascontiguousarray(data)
        h, w, c = data.shape
        if data.dtype == np.uint8:
            internal, fmt, kind = {1: ("R8", "RED"), 2: ("RG8", "RG"), 3: ("RGB8", "RGB"), 4: ("RGBA8", "RGBA")}[c] + ("UNSIGNED_BYTE",)
        else:
            internal, fmt, kind = {1: ("R32F", "RED"), 2: ("RG32F", "RG"), 4: ("RGBA32F", "RGBA")}[c] + ("FLOAT",)
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

    # draw -------------------------------------------------------------------------------------------------
    def draw(self, program: int, width: int, height: int, uniforms: dict, textures: dict, attributes: dict, integers: dict = None,
             region: tuple = None) -> np.ndarray:
        """Fullscreen triangle strip into an RGBA8 target; returns (h, w, 4) uint8, row 0 = bottom.
        uniforms: name → float or tuple of floats; textures: name → texture handle; attributes: name → (4, k) float32;
        integers: name → int for uniforms declared int/bool; region = (x, y, w, h): shade and return that rectangle of the target
        only (scissor test) — bands of frames too large to store"""
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
        for name, value in uniforms.items():
            location = gl.glGetUniformLocation(program, name.encode())
            if location < 0:
                continue
            values = [float(v) for v in (value if hasattr(value, "__len__") else [value])]
            getattr(gl, f"glUniform{len(values)}f")(location, *values)
        for name, value in (integers or {}).items():
            location = gl.glGetUniformLocation(program, name.encode())
            if location >= 0:
                gl.glUniform1i(location, int(value))
        for unit, (name, handle) in enumerate(textures.items()):
            location = gl.glGetUniformLocation(program, name.encode())
            if location < 0:
                continue
            gl.glActiveTexture(GL["TEXTURE0"] + unit)
            gl.glBindTexture(GL["TEXTURE_2D"], handle)
            gl.glUniform1i(location, unit)
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
            gl.glEnable(0x0C11)                                  # GL_SCISSOR_TEST
            gl.glScissor(x, y, rw, rh)
        gl.glDrawArrays(GL["TRIANGLE_STRIP"], 0, 4)
        if region is not None:
            gl.glDisable(0x0C11)
        out = np.zeros((rh, rw, 4), np.uint8)
        gl.glReadPixels(x, y, rw, rh, GL["RGBA"], GL["UNSIGNED_BYTE"], out.ctypes.data)
        assert gl.glGetError() == 0, "GL error after draw"
        gl.glDeleteFramebuffers(1, C.byref(fbo))
        gl.glDeleteTextures(1, C.byref(target))
        return out


if __name__ == "__main__":
    ctx = Context()
    print(ctx.version, "|", ctx.renderer)
    print("float linear:", "GL_OES_texture_float_linear" in ctx.extensions, "| color_buffer_float:", "GL_EXT_color_buffer_float" in ctx.extensions)
    vs = "#version 300 es\nin vec2 p; out vec2 uv; void main(){ gl_Position = vec4(p, 0.0, 1.0); uv = (p + 1.0)/2.0; }"
    fs = "#version 300 es\nprecision highp float; in vec2 uv; out vec4 c; uniform sampler2D t; void main(){ c = texture(t, uv); }"
    prog = ctx.program(vs, fs)
    tex = ctx.texture(np.arange(4*4*3, dtype=np.uint8).reshape(4, 4, 3)*5, True, True, True)
    quad = np.array([[-1, -1], [-1, 1], [1, -1], [1, 1]], np.float32)
    img = ctx.draw(prog, 8, 8, {}, {"t": tex}, {"p": quad})
    print(img[:2, :4, :3].tolist())
