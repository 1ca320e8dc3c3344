"""
ShaderWaveform (reference: shaderflow/audio/waveform.py:14-90): the last `length` seconds of audio reduced to
`length*samplerate` points per channel and written to a (points x 1) RG32F texture every frame. The slicing
arithmetic (`chunk_size`, `_offset = tell % chunk_size`, window ending one sample before the newest, :65-84) is
kept; the reduction itself (`sqrt(mean|x|)` and the RMS/STD variants, :15-22) runs on the device.
"""
from __future__ import annotations

import ctypes as C
import math
from collections.abc import Iterable
from enum import Enum

import numpy as np
from attrs import define

from shaderflow_amd import _native as N
from shaderflow_amd.audio.module import BrokenAudio
from shaderflow_amd.module import ShaderModule
from shaderflow_amd.texture import ShaderTexture
from shaderflow_amd.variable import ShaderVariable, Uniform


class WaveformReducer(Enum):
    Average = 0     # sqrt(mean(|x|))
    RMS = 1         # sqrt(sqrt(mean(x²))·√2)
    STD = 2         # sqrt(std(x))


@define(eq=False, slots=False)
class ShaderWaveform(ShaderModule):
    name: str = "iWaveform"
    audio: BrokenAudio = None
    length: float = 3
    samplerate: float = 60
    reducer: WaveformReducer = WaveformReducer.Average
    smooth: bool = True
    texture: ShaderTexture = None

    @property
    def length_samples(self) -> int:
        return int(max(1, self.length*self.scene.fps))

    def build(self):
        self.texture = ShaderTexture(
            scene=self.scene, filter=("linear" if self.smooth else "nearest"), components=self.audio.channels,
            name=self.name, width=self._points, height=1, mipmaps=False, dtype=np.float32,
        ).repeat(False)

    @property
    def chunk_size(self) -> int:
        return max(1, int(self.length*self.audio.samplerate/self._points))

    @property
    def _points(self) -> int:
        return int(self.length*self.samplerate)

    @property
    def _offset(self) -> int:
        return self.audio.tell % self.chunk_size

    @property
    def _cutoff(self) -> int:
        return int(self.chunk_size*math.floor(self.audio.buffer_size/self.chunk_size))

    def rows(self, tells) -> np.ndarray:
        """(frames, points, channels) float32 for the given `tell`s, computed on the device"""
        native = getattr(self.audio, "native", None)
        if native is None:
            raise RuntimeError("The waveform's audio has no device-resident PCM: load a file into ShaderAudio first")
        tells = np.ascontiguousarray(tells, np.int64)
        out = np.zeros((len(tells), self._points, self.audio.channels), np.float32)
        reducer = WaveformReducer(self.reducer).value if not callable(self.reducer) else 0
        N.check(N.lib().sfx_waveform_rows(native, N.as_ptr(tells, C.c_int64), len(tells), self.chunk_size,
                                          self._points, reducer, N.as_ptr(out, C.c_float)))
        return out

    def update(self):
        if self.texture.components != self.audio.channels:
            self.texture.components = self.audio.channels
        self.texture.write(self.rows([self.audio.tell])[0])

    def pipeline(self) -> Iterable[ShaderVariable]:
        yield Uniform("int", f"{self.name}Length", self.length_samples)
