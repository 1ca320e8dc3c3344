"""
ORACLE (test infrastructure, never shipped, never on the product path).

ctypes binding of oracle/liboracle.so for tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. Nothing under shaderflow_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE/"liboracle.so"


def build(force: bool = False) -> Path:
    sources = [HERE/n for n in ("sfo_audio.c", "sfo_pixel.c", "sfo.h", "sfo_math.h", "Makefile")]
    stale = (not LIB_PATH.exists()) or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in sources)
    if force or stale:
        subprocess.run(["make", "-C", str(HERE), "-B", "liboracle.so"], check=True, capture_output=True)
    return LIB_PATH


class Texture(C.Structure):
    _fields_ = [
        ("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("components", C.c_int32),
        ("dtype", C.c_int32), ("filter", C.c_int32), ("repeat_x", C.c_int32), ("repeat_y", C.c_int32),
        ("levels", C.c_int32), ("mips", C.c_void_p),
    ]


class Uniforms(C.Structure):
    _fields_ = [
        ("iTime", C.c_float), ("iTau", C.c_float), ("iDuration", C.c_float), ("iDeltatime", C.c_float),
        ("iResolution", C.c_float*2),
        ("iWantAspect", C.c_float), ("iQuality", C.c_float), ("iSSAA", C.c_float), ("iFramerate", C.c_float),
        ("iFrame", C.c_int32), ("iRealtime", C.c_int32), ("iLayer", C.c_int32), ("iSubsample", C.c_int32),
        ("iMouse", C.c_float*2),
        ("iMouseInside", C.c_int32), ("iMouse1", C.c_int32), ("iMouse2", C.c_int32),
        ("iCameraMode", C.c_int32), ("iCameraProjection", C.c_int32),
        ("iCameraRight", C.c_float*3), ("iCameraUpward", C.c_float*3), ("iCameraForward", C.c_float*3),
        ("iCameraPosition", C.c_float*3), ("iCameraZenith", C.c_float*3),
        ("iCameraSeparation", C.c_float), ("iCameraZoom", C.c_float), ("iCameraIsometric", C.c_float),
        ("iCameraFocalLength", C.c_float), ("iCameraOrbital", C.c_float), ("iCameraDolly", C.c_float),
        ("iAudioVolume", C.c_float), ("iAudioVolumeIntegral", C.c_float), ("iAudioSTD", C.c_float),
        ("iSpectrogramLength", C.c_int32), ("iSpectrogramBins", C.c_int32),
        ("iSpectrogramSmooth", C.c_int32), ("iSpectrogramScroll", C.c_int32),
        ("iSpectrogramOffset", C.c_float), ("iSpectrogramMin", C.c_float), ("iSpectrogramMax", C.c_float),
        ("iWaveformLength", C.c_int32),
        ("user", C.c_float*16),
    ]


class DynParams(C.Structure):
    _fields_ = [("frequency", C.c_double), ("zeta", C.c_double), ("response", C.c_double),
                ("precision", C.c_double), ("integrate", C.c_int)]


FRAGMENTS = dict(default=0, missing=1, visualizer=2, bars=3, waveform=4, multi_child=5, multi_main=6,
                 shadertoy=7, dynamics=8, audio=9, multipass=10, motionblur=11, life_simulation=12, life_visuals=13,
                 video=14, raymarch=15, mandelbrot=16, tetration=17)
TEX_SLOTS = dict(background=0, iSpectrogram=1, iWaveform=2, child=3)
TEX_HISTORY, TEX_HISTORY_DEPTH, TEX_SLOT_COUNT = 4, 12, 16
DTYPES = {np.dtype(np.uint8): 0, np.dtype(np.float32): 1, np.dtype(np.uint16): 2, np.dtype(np.float16): 3}
MATH_FN = dict(sin=0, cos=1, atan2=2, atan=3, log2=4, exp2=5, pow=6, exp=7, mod=8, smoothstep=9, mix=10, sqrt=11, log=12)

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(LIB_PATH))
        P = C.POINTER
        L.sfo_clock.argtypes = [C.c_double, C.c_double, C.c_int, P(C.c_double), P(C.c_double), P(C.c_double)]
        L.sfo_reader.argtypes = [P(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int64, P(C.c_int32), P(C.c_int64)]
        L.sfo_window.argtypes = [C.c_int, C.c_int, P(C.c_double)]
        L.sfo_fft_power.argtypes = [P(C.c_float), C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, P(C.c_float)]
        L.sfo_fft_amplitude.argtypes = L.sfo_fft_power.argtypes
        L.sfo_filterbank.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double,
                                     P(C.c_int32), P(C.c_int32), P(C.c_float), C.c_int]
        L.sfo_filterbank.restype = C.c_int
        L.sfo_csr_dot.argtypes = [P(C.c_int32), P(C.c_int32), P(C.c_float), C.c_int, P(C.c_float), C.c_int, C.c_int, P(C.c_float)]
        L.sfo_note_of_frequency.argtypes = [C.c_double, C.c_double]
        L.sfo_note_of_frequency.restype = C.c_int
        L.sfo_frequency_of_note.argtypes = [C.c_int, C.c_double]
        L.sfo_frequency_of_note.restype = C.c_double
        L.sfo_from_notes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, P(C.c_double), P(C.c_double), P(C.c_int)]
        L.sfo_dyn_coeffs.argtypes = [P(DynParams), C.c_double, P(C.c_double), P(C.c_double), P(C.c_double)]
        L.sfo_dyn_coeffs.restype = C.c_int
        L.sfo_dyn_step_f32.argtypes = [P(DynParams), C.c_int, P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_float), C.c_double]
        L.sfo_dyn_step_f64.argtypes = [P(DynParams), P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_double), C.c_double, C.c_double]
        L.sfo_waveform_row.argtypes = [P(C.c_float), C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, P(C.c_float)]
        L.sfo_volume_std.argtypes = [P(C.c_float), C.c_int64, C.c_int, C.c_int64, C.c_int, P(C.c_float), P(C.c_float)]
        L.sfo_render.argtypes = [C.c_int, P(Uniforms), P(Texture), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_uint8)]
        L.sfo_render_to.argtypes = [C.c_int, P(Uniforms), P(Texture), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.sfo_resolve.argtypes = [P(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_uint8)]
        L.sfo_sample.argtypes = [P(Texture), C.c_float, C.c_float, P(C.c_float)]
        L.sfo_set_llvmpipe_filter.argtypes = [C.c_int]
        L.sfo_rgb_to_yuv420.argtypes = [P(C.c_uint8), C.c_int, C.c_int, C.c_int, P(C.c_uint8)]
        L.sfo_sample_quad.argtypes = [P(Texture)] + [C.c_float]*6 + [P(C.c_float)]
        L.sfo_mip_levels.argtypes = [C.c_int, C.c_int]
        L.sfo_mip_levels.restype = C.c_int
        L.sfo_mip_offset.argtypes = [P(Texture), C.c_int]
        L.sfo_mip_offset.restype = C.c_int64
        L.sfo_build_mipmaps.argtypes = [P(Texture), C.c_void_p]
        L.sfo_test_math.argtypes = [C.c_int, C.c_float, C.c_float]
        L.sfo_test_math.restype = C.c_float
        _lib = L
    return _lib


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))

# ------------------------------------------------------------------------------------------------ #
# Audio half

def clock(fps: float, frames: int, speed: float = 1.0):
    t = np.zeros(frames); dt = np.zeros(frames); rdt = np.zeros(frames)
    lib().sfo_clock(fps, speed, frames, _p(t, C.c_double), _p(dt, C.c_double), _p(rdt, C.c_double))
    return t, dt, rdt


def reader(rdt: np.ndarray, samplerate: int, channels: int, total_samples: int):
    rdt = np.ascontiguousarray(rdt, np.float64)
    lengths = np.zeros(len(rdt), np.int32); tell = np.zeros(len(rdt), np.int64)
    lib().sfo_reader(_p(rdt, C.c_double), len(rdt), samplerate, channels, total_samples, _p(lengths, C.c_int32), _p(tell, C.c_int64))
    return lengths, tell


def window(kind: int, n: int) -> np.ndarray:
    out = np.zeros(n)
    lib().sfo_window(kind, n, _p(out, C.c_double))
    return out


def fft_power(pcm: np.ndarray, tell: int, fft_n: int = 12, window_kind: int = 0, amplitude: bool = False) -> np.ndarray:
    """pcm: planar (channels, total) float32; amplitude: FourierMagnitude.Amplitude instead of .Power"""
    pcm = np.ascontiguousarray(pcm, np.float32)
    channels, total = pcm.shape
    out = np.zeros((channels, (1 << fft_n)//2 + 1), np.float32)
    fn = lib().sfo_fft_amplitude if amplitude else lib().sfo_fft_power
    fn(_p(pcm, C.c_float), total, channels, tell, fft_n, window_kind, _p(out, C.c_float))
    return out


def resample_linear(data: np.ndarray, ratio: float, n_out: int) -> np.ndarray:
    """samplerate.resample(x, ratio, 'linear') of one channel (spectrogram.py:167; parity unpinned: sfo_audio.c says why)"""
    data = np.ascontiguousarray(data, np.float32)
    out = np.zeros(n_out, np.float32)
    lib().sfo_resample_linear.restype = C.c_int
    lib().sfo_resample_linear.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_double, C.POINTER(C.c_float), C.c_int]
    generated = lib().sfo_resample_linear(_p(data, C.c_float), len(data), float(ratio), _p(out, C.c_float), n_out)
    return out[:generated]


def filterbank(scale: int, interp: int, fmin: float, fmax: float, bins: int, fft_n: int, samplerate: float):
    fft_bins = (1 << fft_n)//2 + 1
    cap = max(64, bins*min(fft_bins, 4096))
    cap = min(cap, bins*fft_bins)
    indptr = np.zeros(bins + 1, np.int32); indices = np.zeros(cap, np.int32); data = np.zeros(cap, np.float32)
    nnz = lib().sfo_filterbank(scale, interp, fmin, fmax, bins, fft_n, samplerate,
                               _p(indptr, C.c_int32), _p(indices, C.c_int32), _p(data, C.c_float), cap)
    assert nnz >= 0
    return indptr, indices[:nnz].copy(), data[:nnz].copy()


def csr_dot(indptr, indices, data, power: np.ndarray) -> np.ndarray:
    power = np.ascontiguousarray(power, np.float32)
    channels, fft_bins = power.shape
    bins = len(indptr) - 1
    indptr = np.ascontiguousarray(indptr, np.int32); indices = np.ascontiguousarray(indices, np.int32)
    data = np.ascontiguousarray(data, np.float32)
    out = np.zeros((bins, channels), np.float32)
    lib().sfo_csr_dot(_p(indptr, C.c_int32), _p(indices, C.c_int32), _p(data, C.c_float), bins,
                      _p(power, C.c_float), channels, fft_bins, _p(out, C.c_float))
    return out


def from_notes(start_note: int, end_note: int, piano: bool, bins: int = 1000, tuning: float = 440.0):
    fmin = C.c_double(); fmax = C.c_double(); b = C.c_int()
    lib().sfo_from_notes(start_note, end_note, int(piano), bins, tuning, C.byref(fmin), C.byref(fmax), C.byref(b))
    return fmin.value, fmax.value, b.value


class DynF32:
    """float32 array DynamicNumber (dynamics.py:77-255)"""
    def __init__(self, n: int, frequency, zeta, response, integrate=False, precision=1e-6):
        self.p = DynParams(frequency, zeta, response, precision, int(integrate))
        self.value = np.zeros(n, np.float32); self.derivative = np.zeros(n, np.float32)
        self.previous = np.zeros(n, np.float32); self.integral = np.zeros(n, np.float32)

    def step(self, target: np.ndarray, dt: float) -> np.ndarray:
        target = np.ascontiguousarray(target, np.float32).ravel()
        lib().sfo_dyn_step_f32(C.byref(self.p), len(self.value), _p(self.value, C.c_float), _p(self.derivative, C.c_float),
                               _p(self.previous, C.c_float), _p(self.integral, C.c_float), _p(target, C.c_float), dt)
        return self.value


class DynF64:
    """float64 scalar DynamicNumber"""
    def __init__(self, value, frequency, zeta, response, integrate=False, precision=1e-6):
        self.p = DynParams(frequency, zeta, response, precision, int(integrate))
        self.value = C.c_double(value); self.derivative = C.c_double(0.0)
        self.previous = C.c_double(value); self.integral = C.c_double(0.0)

    def step(self, target: float, dt: float) -> float:
        lib().sfo_dyn_step_f64(C.byref(self.p), C.byref(self.value), C.byref(self.derivative),
                               C.byref(self.previous), C.byref(self.integral), float(target), dt)
        return self.value.value


def waveform_row(pcm: np.ndarray, tell: int, chunk_size: int, points: int, reducer: int = 0) -> np.ndarray:
    pcm = np.ascontiguousarray(pcm, np.float32)
    channels, total = pcm.shape
    out = np.zeros((points, channels), np.float32)
    lib().sfo_waveform_row(_p(pcm, C.c_float), total, channels, tell, chunk_size, points, reducer, _p(out, C.c_float))
    return out


def volume_std(pcm: np.ndarray, tell: int, n: int):
    pcm = np.ascontiguousarray(pcm, np.float32)
    channels, total = pcm.shape
    vol = C.c_float(); std = C.c_float()
    lib().sfo_volume_std(_p(pcm, C.c_float), total, channels, tell, n, C.byref(vol), C.byref(std))
    return vol.value, std.value

# ------------------------------------------------------------------------------------------------ #
# Pixel half

def make_texture(data: np.ndarray, filter: str = "linear", repeat_x: bool = True, repeat_y: bool = True) -> Texture:
    """data: (height, width, components) with row 0 = bottom (GL order). Keeps a reference."""
    data = np.ascontiguousarray(data)
    if data.ndim == 2:
        data = data[:, :, None]
    t = Texture(data.ctypes.data, data.shape[1], data.shape[0], data.shape[2], DTYPES[data.dtype],
                1 if filter == "linear" else 0, int(repeat_x), int(repeat_y), 0, None)
    t._keep = data
    return t


def build_mipmaps(t: Texture, source: np.ndarray | None = None) -> Texture:
    """texture.build_mipmaps() + the mipmap minification filter (texture.py:131-137, 277-278): the chain is built from `source`
    (default: the texture's own level 0 — pass the content level 0 had when the reference last called apply(), e.g. zeros for a
    texture that was only ever filled by from_numpy) and attached; t.filter becomes LINEAR_MIPMAP / NEAREST_MIPMAP"""
    levels = lib().sfo_mip_levels(t.width, t.height)
    whole = Texture.from_buffer_copy(t)
    whole.levels = levels
    chain = np.zeros(max(16, lib().sfo_mip_offset(C.byref(whole), levels)), np.uint8)
    base = whole
    if source is not None:
        source = np.ascontiguousarray(source)
        base = Texture.from_buffer_copy(t)
        base.data = source.ctypes.data
    lib().sfo_build_mipmaps(C.byref(base), chain.ctypes.data)
    t.levels, t.mips, t.filter = levels, chain.ctypes.data, (2 if (t.filter & 1) else 3)
    t._chain = chain
    return t


def mip_level(t: Texture, level: int) -> np.ndarray:
    """level >= 1 of a built chain as (h, w, components)"""
    w, h = max(1, t.width >> level), max(1, t.height >> level)
    dtype = {v: k for k, v in DTYPES.items()}[t.dtype]
    offset = lib().sfo_mip_offset(C.byref(t), level)
    return np.frombuffer(t._chain, dtype, count=w*h*t.components, offset=offset).reshape(h, w, t.components).copy()


def default_uniforms(width: int, height: int, **kw) -> Uniforms:
    """Defaults of a freshly built scene (scene.py:687-703, camera.py:147-185,196-201)"""
    u = Uniforms()
    u.iResolution[0] = width; u.iResolution[1] = height
    u.iWantAspect = width/height
    u.iQuality = 0.5; u.iSSAA = 1.0; u.iFramerate = 60.0; u.iDuration = 10.0
    u.iSubsample = 2
    u.iCameraMode = 1; u.iCameraProjection = 0
    u.iCameraRight[0] = 1.0; u.iCameraUpward[1] = 1.0; u.iCameraForward[2] = 1.0
    u.iCameraZenith[1] = 1.0
    u.iCameraSeparation = 0.05; u.iCameraZoom = 1.0; u.iCameraFocalLength = 1.0
    for key, value in kw.items():
        cur = getattr(u, key)
        if hasattr(cur, "__len__"):
            for i, v in enumerate(value):
                cur[i] = v
        else:
            setattr(u, key, value)
    return u


def _slots(textures: dict) -> "C.Array":
    """name → texture; a key ("history", t) (or an int t) is the history slot `<name>{t}x0`"""
    slots = (Texture*TEX_SLOT_COUNT)()
    for name, tex in textures.items():
        if isinstance(name, tuple):
            name = name[1]
        slots[TEX_HISTORY + name if isinstance(name, int) else TEX_SLOTS[name]] = tex
    return slots


def render(fragment: str, u: Uniforms, textures: dict, wr: int, hr: int,
           rows: tuple[int, int] | None = None, threads: int = 1) -> np.ndarray:
    """Returns (hr, wr, 4) uint8, row 0 = bottom; rows outside `rows` stay zero"""
    out = np.zeros((hr, wr, 4), np.uint8)
    y0, y1 = rows or (0, hr)
    lib().sfo_render(FRAGMENTS[fragment], C.byref(u), _slots(textures), wr, hr, y0, y1, threads, _p(out, C.c_uint8))
    return out


def render_to(fragment: str, u: Uniforms, textures: dict, wr: int, hr: int, components: int, dtype,
              threads: int = 1) -> np.ndarray:
    """Render into a (hr, wr, components) target of uint8 or float32 (any ShaderTexture format, texture.py:177-184)"""
    out = np.zeros((hr, wr, components), dtype)
    lib().sfo_render_to(FRAGMENTS[fragment], C.byref(u), _slots(textures), wr, hr, 0, hr, threads, components,
                        DTYPES[np.dtype(dtype)], out.ctypes.data_as(C.c_void_p))
    return out


def resolve(screen: np.ndarray, w: int, h: int, subsample: int,
            rows: tuple[int, int] | None = None, threads: int = 1) -> np.ndarray:
    screen = np.ascontiguousarray(screen, np.uint8)
    hr, wr = screen.shape[:2]
    out = np.zeros((h, w, 3), np.uint8)
    y0, y1 = rows or (0, h)
    lib().sfo_resolve(_p(screen, C.c_uint8), wr, hr, w, h, subsample, y0, y1, threads, _p(out, C.c_uint8))
    return out


def rgb_to_yuv420(frame: np.ndarray, matrix: int = 0) -> np.ndarray:
    """(h, w, 3) uint8 → the w*h*3/2 bytes of planar yuv420p, rows in the frame's own order"""
    frame = np.ascontiguousarray(frame, np.uint8)
    h, w = frame.shape[:2]
    out = np.zeros(w*h*3//2, np.uint8)
    lib().sfo_rgb_to_yuv420(_p(frame, C.c_uint8), w, h, matrix, _p(out, C.c_uint8))
    return out


def sample_quad(tex: Texture, here, right, above) -> np.ndarray:
    """texture() at `here` with the implicit derivatives a 2x2 quad provides: `right` / `above` are the coordinates of the pixel's
    horizontal and vertical quad neighbours (sfo_sample_quad)"""
    out = np.zeros(4, np.float32)
    lib().sfo_sample_quad(C.byref(tex), here[0], here[1], right[0], right[1], above[0], above[1], _p(out, C.c_float))
    return out


class llvmpipe_filter:
    """`with O.llvmpipe_filter():` — unorm8 textures are filtered as Mesa llvmpipe filters them (24.8 fixed-point coordinates, 8-bit
    weights, every lerp rounded to 8 bits: sfo_pixel.c). A checker's switch for the tests that demonstrate where the > 1 LSB values
    against the llvmpipe goldens come from; off by default and for every parity test of the HIP kernels."""

    def __enter__(self):
        lib().sfo_set_llvmpipe_filter(1)
        return self

    def __exit__(self, *exc):
        lib().sfo_set_llvmpipe_filter(0)
        return False


def sample(tex: Texture, s: float, t: float) -> np.ndarray:
    out = np.zeros(4, np.float32)
    lib().sfo_sample(C.byref(tex), s, t, _p(out, C.c_float))
    return out


def math(fn: str, a, b=0.0) -> np.ndarray:
    a = np.asarray(a, np.float32); b = np.broadcast_to(np.asarray(b, np.float32), a.shape)
    f = lib().sfo_test_math
    code = MATH_FN[fn]
    return np.array([f(code, float(x), float(y)) for x, y in zip(a.ravel(), b.ravel())], np.float32).reshape(a.shape)
