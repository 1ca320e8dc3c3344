"""
BrokenSpectrogram / ShaderSpectrogram (reference: shaderflow/audio/spectrogram.py:20-320).

Host side: the option namespaces (FourierMagnitude, FourierVolume, SpectrogramInterpolation, SpectrogramScale,
SpectrogramWindow) and the filterbank construction (`spectrogram_frequencies`, `spectrogram_matrix`, `from_notes`)
stay numpy/scipy code — they run once per configuration and define WHAT is computed. Device side: the per-frame
STFT (`window*frame → radix-2 FFT in float64 → |X|² → float32`, :155-171) and the filterbank product
(:175-176) are HIP kernels reached through an `sfx_stft_plan`. Options the kernels implement: the three windows,
`FourierMagnitude.Power` and `.Amplitude` (any other `magnitude` callable runs on the host between the device's FFT and filterbank:
`device_magnitude`), any scale/interpolation (they only shape the CSR matrix). `sample_rateio != 1`
(:158-167 resamples with the third-party `samplerate` package's 'linear' converter — un-vendored, not importable here): the converter's
read positions depend on the sizes and the ratio alone, so its float64 position loop runs here once per plan (`linear_resample_taps`)
and the device interpolates; PARITY UNPINNED for that option (no reference output can be generated without the package: the
restatement of libsamplerate's src_linear.c is checked against the oracle's own sequential restatement and a hand-computed vector).
"""
from __future__ import annotations

import ctypes as C
import functools
import math
from collections.abc import Callable, Iterable
from typing import Union

import numpy as np
from attrs import Factory, define, field

from shaderflow_amd import _native as N
from shaderflow_amd.audio.module import BrokenAudio
from shaderflow_amd.dynamics import DynamicNumber
from shaderflow_amd.module import ShaderModule
from shaderflow_amd.piano.notes import PianoNote
from shaderflow_amd.texture import ShaderTexture
from shaderflow_amd.variable import ShaderVariable, Uniform


class FourierMagnitude:
    def Amplitude(x: np.ndarray) -> np.ndarray:
        return np.abs(x)

    def Power(x: np.ndarray) -> np.ndarray:
        return (x*x.conjugate()).real


class FourierVolume:
    def dBFS(x): return 10*np.log10(x)
    def Sqrt(x): return np.sqrt(x)
    def Linear(x): return x
    def dBFsTremx(x): return 10*(np.log10(x + 0.1) + 1)/1.0414


class SpectrogramInterpolation:
    """Discrete FFT bins → continuous frequency: one kernel per filterbank row (spectrogram.py:44-70)"""

    def make_euler(end: float = 1.54) -> Callable:
        return (lambda x: np.exp(-(2*x/end)**2)/(end*(math.pi**0.5)))

    def Dirac(x):
        dirac = np.zeros(x.shape)
        dirac[np.round(x) == 0] = 1
        return dirac

    Euler = make_euler(end=1.2)

    def Sinc(x: np.ndarray) -> np.ndarray:
        return np.abs(np.sinc(x))


class SpectrogramScale:
    """(forward, inverse) of the vertical axis"""
    Octave = ((lambda x: (np.log(x)/np.log(2))), (lambda x: (2**x)))
    MEL = ((lambda x: 2595*np.log10(1 + x/700)), (lambda x: 700*(10**(x/2595) - 1)))


class SpectrogramWindow:

    @functools.lru_cache
    def hann_poisson_window(N: int, alpha: float = 2.0) -> np.ndarray:
        n = np.arange(N)
        return (0.5*(1 - np.cos(2*np.pi*n/N)))*np.exp(-alpha*np.abs(N - 2*n)/N)

    @functools.lru_cache
    def hanning(size: int) -> np.ndarray:
        return np.hanning(size)

    @functools.lru_cache
    def none(size: int) -> np.ndarray:
        return np.ones(size)


def linear_resample_taps(n_in: int, ratio: float, n_out: int) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Where `samplerate.resample(x, ratio, 'linear')` (spectrogram.py:167) reads: output n = float32(x[a[n]] + w[n]*(x[b[n]] - x[a[n]])),
    evaluated in float64. libsamplerate's linear converter (src_linear.c, through src_simple: a fresh state whose `last_value` is x[0])
    emits x[0] while its float64 `input_index` is below one, then interpolates between x[in_used - 1] and x[in_used] — a one-sample
    delay — advancing input_index += 1/ratio and carrying the integer part into `in_used` with fmod_one / lrint after every output.
    Nothing of it depends on the samples, so the loop runs here. Returns as many taps as the converter generates (<= n_out)."""
    def fmod_one(x: float) -> float:
        res = x - float(np.rint(x))                                  # lrint: round half to even
        return res + 1.0 if res < 0.0 else res
    a, b, w = np.zeros(n_out, np.int32), np.zeros(n_out, np.int32), np.zeros(n_out, np.float64)
    out, input_index, in_used, step = 0, 0.0, 0, 1.0/float(ratio)
    while input_index < 1.0 and out < n_out:
        if in_used + (1.0 + input_index) >= n_in:
            break
        a[out], b[out], w[out] = 0, 0, input_index                   # last_value + input_index*(x[0] - last_value) with last_value = x[0]
        out += 1
        input_index += step
    rem = fmod_one(input_index)
    in_used += int(np.rint(input_index - rem))
    input_index = rem
    while out < n_out and in_used + input_index < n_in:
        a[out], b[out], w[out] = in_used - 1, in_used, input_index
        out += 1
        input_index += step
        rem = fmod_one(input_index)
        in_used += int(np.rint(input_index - rem))
        input_index = rem
    return a[:out], b[:out], w[:out]


_WINDOW_CODES = {SpectrogramWindow.hanning: 0, SpectrogramWindow.hann_poisson_window: 1, SpectrogramWindow.none: 2}


@define(eq=False, slots=False)
class BrokenSpectrogram:
    audio: BrokenAudio = Factory(BrokenAudio)
    fft_n: int = field(default=12, converter=int)
    sample_rateio: int = field(default=1, converter=int)
    scale: tuple = SpectrogramScale.Octave
    interpolation: Callable = SpectrogramInterpolation.Euler
    magnitude: Callable = FourierMagnitude.Power
    window: Callable = SpectrogramWindow.hanning
    volume: Callable = FourierVolume.Sqrt

    minimum_frequency: float = 20.0
    maximum_frequency: float = 20000.0
    spectrogram_bins: int = 1000

    def __hash__(self) -> int:
        return hash((self.fft_n, self.minimum_frequency, self.maximum_frequency, self.spectrogram_bins,
                     self.sample_rateio, self.magnitude, self.interpolation, self.scale, self.volume))

    @property
    def fft_size(self) -> int:
        return int(2**(self.fft_n)*self.sample_rateio)

    @property
    def fft_bins(self) -> int:
        return int(self.fft_size/2 + 1)

    @property
    def fft_frequencies(self) -> np.ndarray:
        return np.fft.rfftfreq(self.fft_size, 1/(self.audio.samplerate*self.sample_rateio))

    @property
    def spectrogram_frequencies(self) -> np.ndarray:
        return self.scale[1](np.linspace(self.scale[0](self.minimum_frequency), self.scale[0](self.maximum_frequency), self.spectrogram_bins))

    @functools.lru_cache
    def spectrogram_matrix(self):
        """(bins, fft_bins) float32 filterbank as scipy CSR: row r is the interpolation kernel centred on
        frequency r measured in FFT bins, entries below 1e-5 dropped (spectrogram.py:194-224)"""
        import scipy.sparse
        centres = self.spectrogram_frequencies/self.fft_frequencies[1]
        matrix = np.array([self.interpolation(index - np.arange(self.fft_bins)) for index in centres], dtype=self.audio.dtype)
        matrix[np.abs(matrix) < 1e-5] = 0
        return scipy.sparse.csr_matrix(matrix)

    def from_notes(self, start, end, bins: int = 1000, piano: bool = False, tuning: float = 440):
        start = PianoNote.get(start, tuning=tuning)
        end = PianoNote.get(end, tuning=tuning)
        self.minimum_frequency = start.frequency
        self.maximum_frequency = end.frequency
        if not piano:
            self.spectrogram_bins = bins
        else:
            half_semitone = 2**(0.5/12)                       # one bin per key, edges half a semitone out
            self.spectrogram_bins = ((end.note - start.note) + 1)
            self.minimum_frequency /= half_semitone
            self.maximum_frequency *= half_semitone

    # device plan ----------------------------------------------------------------------------------------------

    _plan: N.Handle = None
    _plan_key: tuple = None

    def _context(self) -> N.Context:
        scene = getattr(self, "scene", None)
        return scene.context if scene is not None else N.default_context()

    def plan(self) -> N.Handle:
        """The sfx_stft_plan of the current configuration (rebuilt when it changes)"""
        taps = None
        if self.sample_rateio != 1:
            # the converter must deliver exactly fft_size samples, or `window(fft_size) * data` does not broadcast (spectrogram.py:169-170)
            taps = linear_resample_taps(int(2**self.fft_n), self.sample_rateio, self.fft_size)
            if self.sample_rateio < 1 or len(taps[0]) != self.fft_size:
                raise ValueError(f"operands could not be broadcast together with shapes ({self.fft_size},) ({self.audio.channels},{len(taps[0])}) "
                                 f"(sample_rateio = {self.sample_rateio})")
        if not callable(self.magnitude):
            raise TypeError(f"spectrogram magnitude {self.magnitude!r} is not callable")
        if self.fft_size > 16384:
            # (the reference computes any size; here a frame's transform lives in one workgroup's LDS — 16 384 float64 inputs at most. Said
            # here, in the caller's terms, not only as the library's UNSUPPORTED: ADVICE round 5)
            raise ValueError(f"fft_size = 2**fft_n * sample_rateio = {self.fft_size} samples: the device transform takes at most 16384 "
                             f"(fft_n = {self.fft_n}, sample_rateio = {self.sample_rateio}: lower one of them)")
        custom = None
        if self.window not in _WINDOW_CODES:
            # a window function of the user's own (the reference multiplies by whatever `self.window(N)` returns, in float64,
            # spectrogram.py:155-171): evaluated here like there, its values replace the plan's table
            if not callable(self.window):
                raise TypeError(f"spectrogram window {self.window!r} is neither a SpectrogramWindow nor a callable")
            custom = np.ascontiguousarray(self.window(self.fft_size), np.float64)
            if custom.shape != (self.fft_size,):
                raise ValueError(f"window function returned shape {custom.shape}, expected ({self.fft_size},)")
        amplitude = (self.magnitude is FourierMagnitude.Amplitude)
        key = (hash(self), self.audio.channels, self.audio.samplerate, _WINDOW_CODES.get(self.window, custom.tobytes() if custom is not None else None), amplitude)
        if self._plan is None or self._plan_key != key:
            self.release_plan()
            matrix = self.spectrogram_matrix()
            indptr = np.ascontiguousarray(matrix.indptr, np.int32)
            indices = np.ascontiguousarray(matrix.indices, np.int32)
            data = np.ascontiguousarray(matrix.data, np.float32)
            handle = N.Handle()
            code = _WINDOW_CODES.get(self.window, _WINDOW_CODES[SpectrogramWindow.none])
            if taps is None:
                N.check(N.lib().sfx_stft_plan(self._context().handle, self.fft_n, code, self.spectrogram_bins,
                                              self.audio.channels, N.as_ptr(indptr, C.c_int32), N.as_ptr(indices, C.c_int32),
                                              N.as_ptr(data, C.c_float), C.byref(handle)))
            else:
                tap_a, tap_b, tap_w = (np.ascontiguousarray(t) for t in taps)
                N.check(N.lib().sfx_stft_plan_resampled(self._context().handle, self.fft_n, self.fft_size, N.as_ptr(tap_a, C.c_int32), N.as_ptr(tap_b, C.c_int32),
                                                        N.as_ptr(tap_w, C.c_double), code, self.spectrogram_bins, self.audio.channels,
                                                        N.as_ptr(indptr, C.c_int32), N.as_ptr(indices, C.c_int32), N.as_ptr(data, C.c_float), C.byref(handle)))
            N.check(N.lib().sfx_stft_plan_magnitude(handle, int(amplitude)))
            if custom is not None:
                N.check(N.lib().sfx_stft_plan_window(handle, N.as_ptr(custom, C.c_double), int(custom.size)))
            self._plan, self._plan_key = handle, key
        return self._plan

    def release_plan(self) -> None:
        if self._plan is not None and self._plan.value:
            N.lib().sfx_stft_plan_destroy(self._plan)
        self._plan, self._plan_key = None, None

    def _native_audio(self) -> N.Handle:
        native = getattr(self.audio, "native", None)
        if native is None:
            raise RuntimeError("The spectrogram's audio has no device-resident PCM: load a file into ShaderAudio first")
        return native

    @property
    def device_magnitude(self) -> bool:
        """Whether `magnitude` is one the STFT kernel evaluates itself. Any OTHER callable (the reference takes any function of the complex
        spectrum, spectrogram.py:20-41, 169-171) is applied HERE, on the host, between two device stages: the float64 spectrum comes back
        from the device (sfx_stft_spectrum), the callable runs in numpy exactly as `self.magnitude(np.fft.rfft(…)).astype(dtype)` does there,
        and its result goes through the filterbank on the device (sfx_filterbank_apply) — slow (two round trips per frame), but it works;
        the batched frame tape steps aside for it (tape.FrameTape.applicable)."""
        return self.magnitude in (FourierMagnitude.Power, FourierMagnitude.Amplitude)

    def spectrum(self) -> np.ndarray:
        """(channels, fft_bins) complex128: np.fft.rfft(window*frame) of the last 2**fft_n samples, computed on the device (:169-170)"""
        tell = np.array([self.audio.tell], np.int64)
        pairs = np.zeros((self.audio.channels, self.fft_bins, 2), np.float64)
        N.check(N.lib().sfx_stft_spectrum(self.plan(), self._native_audio(), N.as_ptr(tell, C.c_int64), 1, N.as_ptr(pairs, C.c_double)))
        return pairs.view(np.complex128)[..., 0]

    def fft(self) -> np.ndarray:
        """(channels, fft_bins) float32 power of the last 2**fft_n samples (spectrogram.py:155-171)"""
        if not self.device_magnitude:
            return np.asarray(self.magnitude(self.spectrum())).astype(self.audio.dtype)       # :169-171
        tell = np.array([self.audio.tell], np.int64)
        out = np.zeros((self.audio.channels, self.fft_bins), np.float32)
        N.check(N.lib().sfx_stft_power(self.plan(), self._native_audio(), N.as_ptr(tell, C.c_int64), 1, N.as_ptr(out, C.c_float)))
        return out

    def next(self) -> np.ndarray:
        """spectrogram_matrix().dot(fft().T).T: a (channels, bins) VIEW of a (bins, channels) buffer (:175-176)"""
        out = np.zeros((self.spectrogram_bins, self.audio.channels), np.float32)
        if not self.device_magnitude:
            magnitudes = np.ascontiguousarray(self.fft(), np.float32)
            if magnitudes.shape != (self.audio.channels, self.fft_bins):
                raise ValueError(f"magnitude callable returned shape {magnitudes.shape}, expected ({self.audio.channels}, {self.fft_bins})")
            N.check(N.lib().sfx_filterbank_apply(self.plan(), N.as_ptr(magnitudes, C.c_float), 1, 0, N.as_ptr(out, C.c_float)))
            return out.T
        tell = np.array([self.audio.tell], np.int64)
        N.check(N.lib().sfx_spectrogram_targets(self.plan(), self._native_audio(), N.as_ptr(tell, C.c_int64), 1, 0, N.as_ptr(out, C.c_float)))
        return out.T


@define(eq=False, slots=False)
class ShaderSpectrogram(BrokenSpectrogram, ShaderModule):
    name: str = "iSpectrogram"
    length: float = 5
    offset: int = 0
    smooth: bool = False
    scrolling: bool = False
    dynamics: DynamicNumber = None
    texture: ShaderTexture = None

    @property
    def length_samples(self) -> int:
        return int(max(1, self.length*self.scene.fps))

    @property
    def _row_shape(self) -> tuple[int, int]:
        return (self.audio.channels, self.spectrogram_bins)

    def __attrs_post_init__(self):
        ShaderModule.__attrs_post_init__(self)
        self.dynamics = DynamicNumber(frequency=4, zeta=1, response=0, dtype=np.float32)
        self.texture = ShaderTexture(scene=self.scene, name=self.name, dtype=np.float32, repeat_y=False)

    def configure_texture(self) -> None:
        """Size and sampler state of the scrolling texture (spectrogram.py:299-302)"""
        self.texture.components = self.audio.channels
        self.texture.filter = ("linear" if self.smooth else "nearest")
        self.texture.height = self.spectrogram_bins
        self.texture.width = self.length_samples

    def update(self):
        self.configure_texture()
        self.offset = (self.offset + 1) % self.length_samples
        if (self.dynamics.value.shape != (self._row_shape)):
            self.dynamics.set(np.zeros(self._row_shape, dtype=np.float32))
        # The (bins, 2) buffer re-viewed as (2, bins): bytes stay [bin0_L, bin0_R, bin1_L, …] = RG texels (:306)
        self.dynamics.target = self.next().T.reshape(2, -1)
        self.dynamics.next(dt=abs(self.scene.dt))
        self.texture.write(viewport=(self.offset, 0, 1, self.spectrogram_bins), data=self.dynamics.value.astype(np.float32))

    def pipeline(self) -> Iterable[ShaderVariable]:
        yield Uniform("int", f"{self.name}Length", self.length_samples)
        yield Uniform("int", f"{self.name}Bins", self.spectrogram_bins)
        yield Uniform("float", f"{self.name}Offset", self.offset/self.length_samples)
        yield Uniform("int", f"{self.name}Smooth", self.smooth)
        yield Uniform("float", f"{self.name}Min", self.spectrogram_frequencies[0])
        yield Uniform("float", f"{self.name}Max", self.spectrogram_frequencies[-1])
        yield Uniform("bool", f"{self.name}Scroll", self.scrolling)

    def destroy(self) -> None:
        self.release_plan()
