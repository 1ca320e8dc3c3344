"""Shared helpers of the parity tests: the same inputs are handed to the C-ABI (HIP) and to the oracle."""
from __future__ import annotations

import ctypes as C

import numpy as np

from oracle import binding as O
from shaderflow_amd import _native as N


def i16_to_f32(pcm):
    return (pcm.astype(np.float32)/np.float32(32768.0)).astype(np.float32)


class Gpu:
    """Thin harness over the C-ABI (one context per test session)"""
    _ctx = None

    def __init__(self):
        if Gpu._ctx is None:
            Gpu._ctx = N.Context(0)
        self.ctx = Gpu._ctx
        self.lib = N.lib()
        self._live = []

    def texture(self, data: np.ndarray, filter="linear", repeat_x=True, repeat_y=True) -> N.Handle:
        """data: (h, w, c), row 0 = bottom"""
        data = np.ascontiguousarray(data)
        if data.ndim == 2:
            data = data[:, :, None]
        h = N.Handle()
        N.check(self.lib.sfx_texture_create(self.ctx.handle, data.shape[1], data.shape[0], data.shape[2], N.NUMPY_DTYPES[data.dtype], C.byref(h)))
        N.check(self.lib.sfx_texture_params(h, 1 if filter == "linear" else 0, int(repeat_x), int(repeat_y)))
        N.check(self.lib.sfx_texture_write(h, data.ctypes.data, data.nbytes, 0, 0, 0, 0))
        self._live.append(h)
        return h

    def empty(self, w, h, comps, dtype=np.uint8) -> N.Handle:
        t = N.Handle()
        N.check(self.lib.sfx_texture_create(self.ctx.handle, w, h, comps, N.NUMPY_DTYPES[np.dtype(dtype)], C.byref(t)))
        self._live.append(t)
        return t

    def read(self, tex: N.Handle, w, h, comps, dtype=np.uint8) -> np.ndarray:
        out = np.empty((h, w, comps), dtype)
        N.check(self.lib.sfx_texture_read(tex, out.ctypes.data, out.nbytes))
        return out

    def program(self, source: str):
        p, fb = N.Handle(), C.c_int()
        N.check(self.lib.sfx_program_lookup(self.ctx.handle, source.encode(), C.byref(p), C.byref(fb)))
        return p, bool(fb.value)

    def set_uniforms(self, prog, u: O.Uniforms):
        """Push every field of an oracle uniform block by name through sfx_uniform_set"""
        for name, ctype in u._fields_:
            if name == "user":
                continue
            value = getattr(u, name)
            if hasattr(value, "__len__"):
                arr = np.array(list(value), np.float32)
                code = {2: N.T_VEC2, 3: N.T_VEC3}[len(arr)]
            elif ctype is C.c_int32:
                arr = np.array([value], np.int32); code = N.T_INT
            else:
                arr = np.array([value], np.float32); code = N.T_FLOAT
            N.check(self.lib.sfx_uniform_set(prog, name.encode(), code, arr.ctypes.data, None))

    def set_float(self, prog, name: str, value: float) -> bool:
        arr = np.array([value], np.float32)
        known = C.c_int()
        N.check(self.lib.sfx_uniform_set(prog, name.encode(), N.T_FLOAT, arr.ctypes.data, C.byref(known)))
        return bool(known.value)

    def bind(self, prog, name: str, tex: N.Handle):
        known = C.c_int()
        N.check(self.lib.sfx_sampler_bind(prog, name.encode(), tex, C.byref(known)))
        return bool(known.value)

    def render(self, prog, w, h, comps=4, dtype=np.uint8, layer=0) -> np.ndarray:
        target = self.empty(w, h, comps, dtype)
        N.check(self.lib.sfx_render(prog, target, layer))
        return self.read(target, w, h, comps, dtype)

    def set_values(self, prog, name: str, values, integer=False) -> bool:
        """Push a scalar / vec2-4 uniform by name; True when the program consumes it"""
        arr = np.atleast_1d(np.asarray(values, np.int32 if integer else np.float32))
        code = N.T_INT if integer else {1: N.T_FLOAT, 2: N.T_VEC2, 3: N.T_VEC3, 4: N.T_VEC4}[arr.size]
        known = C.c_int()
        N.check(self.lib.sfx_uniform_set(prog, name.encode(), code, arr.ctypes.data, C.byref(known)))
        return bool(known.value)

    def resolve(self, screen: np.ndarray, w, h, subsample) -> np.ndarray:
        src = self.texture(screen, "linear", False, False)
        dst = self.empty(w, h, 3)
        N.check(self.lib.sfx_resolve(self.ctx.handle, src, dst, subsample))
        return self.read(dst, w, h, 3)

    def render_resolve(self, prog, w, h, ssaa, subsample) -> np.ndarray:
        dst = self.empty(w, h, 3)
        N.check(self.lib.sfx_render_resolve(prog, dst, ssaa, subsample))
        return self.read(dst, w, h, 3)

    def close(self):
        for h in self._live:
            self.lib.sfx_texture_destroy(h)
        self._live.clear()


def usable_cores() -> int:
    """Cores the oracle's thread pool may really use: the CPUs this process may be scheduled on, capped by the cgroup's CPU quota. The pool's
    GPU boxes show 256 logical CPUs and grant 16 CPUs' worth of time: 256 runnable threads on 16 cores ran the oracle at 12x one thread
    where 16 threads give 17.7x (tools/experiments/oracle_scaling.py) — a third of the full-size tests' time"""
    import os
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    quota = None
    try:
        limit, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if limit != "max":
            quota = max(1, int(float(limit)/float(period) + 0.5))
    except (OSError, ValueError):
        try:
            limit, period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if limit > 0:
                quota = max(1, int(limit/period + 0.5))
        except (OSError, ValueError):
            pass
    return max(1, min(visible, quota) if quota else visible)


def visualizer_inputs(w, h, seed=0, bg_size=(96, 54), volume=0.8, std=0.2, time=1.25, bins=115):
    """Random but plausible inputs of visualizer.frag: (oracle uniforms, {name: ndarray}, sampler params)"""
    rng = np.random.default_rng(seed)
    bg = rng.integers(0, 256, (bg_size[1], bg_size[0], 3), dtype=np.uint8)
    spec = (np.abs(rng.standard_normal((bins, 1, 2)))*4000).astype(np.float32)
    wave = np.abs(rng.standard_normal((1, 180, 2))).astype(np.float32)*0.6
    u = O.default_uniforms(w, h, iTime=time, iTau=(time/10.0) % 1.0, iAudioVolume=volume, iAudioVolumeIntegral=volume*time,
                           iAudioSTD=std, iSpectrogramBins=bins, iSpectrogramLength=1, iWaveformLength=180)
    arrays = {"background": bg, "iSpectrogram": spec, "iWaveform": wave}
    params = {"background": ("linear", True, True), "iSpectrogram": ("nearest", True, False), "iWaveform": ("linear", False, False)}
    return u, arrays, params


def smooth_spectrum(bins: int = 115, seed: int = 0) -> np.ndarray:
    """A spectrogram column like music's after the DynamicNumber smoothing: a few broad peaks over a low floor, (bins, 1, 2) float32 —
    neighbouring bars differ little, so most of a frame's wave tiles take the strip kernel's pixel tier (visualizer_inputs' white-noise
    column, whose neighbouring bars differ by whole bar heights, sends nearly every tile to the per-sample path)"""
    rng = np.random.default_rng(seed)
    b = np.arange(bins, dtype=np.float64)[:, None]
    column = np.full((bins, 2), 2.0)
    for _ in range(3):
        centre, width, height = rng.uniform(8, bins - 8, 2), rng.uniform(5.0, 12.0, 2), rng.uniform(300.0, 2500.0, 2)
        column += height*np.exp(-((b - centre)/width)**2)
    return column.reshape(bins, 1, 2).astype(np.float32)


def mip_probe_texture(width: int, height: int, dtype) -> np.ndarray:
    """The texture of the mipmap probes (tests/golden/make_golden_mip.py renders it on the reference): smooth gradients plus seeded
    noise, so that neighbouring levels differ visibly; (height, width, 4), row 0 = bottom"""
    rng = np.random.default_rng(17)
    y, x = np.mgrid[0:height, 0:width]
    base = np.stack([x/(width - 1), y/(height - 1), 0.5 + 0.5*np.sin(x*0.7)*np.cos(y*0.5), np.ones_like(x, float)], axis=-1)
    image = np.clip(0.7*base + 0.3*rng.random((height, width, 4)), 0.0, 1.0)
    if np.dtype(dtype) == np.uint8:
        return np.rint(image*255).astype(np.uint8)
    return image.astype(np.float32)


def oracle_textures(arrays, params):
    return {k: O.make_texture(v, *params[k]) for k, v in arrays.items()}


def gpu_bind_all(gpu: Gpu, prog, arrays, params):
    for k, v in arrays.items():
        assert gpu.bind(prog, k, gpu.texture(v, *params[k]))


def lsb_report(got: np.ndarray, want: np.ndarray) -> str:
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    return f"max {d.max()} LSB, {int((d > 0).sum())}/{d.size} differ, {int((d > 1).sum())} above 1 LSB"


# A fragment of this repository's own for scrolling spectrograms (ShaderSpectrogram(length > 0), spectrogram.py:272-311): the texture is
# `length*fps` columns wide, one column is rewritten per frame, iSpectrogramOffset tells where. Rendered by the reference on Mesa
# (tests/golden/make_golden_mesa.py) and by the product through the run-time translator (tests/test_gpu_translated.py, test_gpu_mesa.py).
SCROLL_FRAGMENT = """
void main() {
    vec2 uv = vec2(astuv.x + iSpectrogramOffset, astuv.y);
    vec2 s = sqrt(texture(iSpectrogram, uv).xy)/40.0;
    fragColor = vec4(s, float(iSpectrogramLength)/64.0, 1);
}
"""
