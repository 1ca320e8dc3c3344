"""
ShaderWaveform — the recent audio history as a one-row texture.

Every frame the last `length` seconds are cut into `length*samplerate` chunks per channel and each chunk is reduced
to one value (default `sqrt(mean|x|)`), giving a `(points × 1)` RG32F texture named `iWaveform` that fragments
sample with `texture(iWaveform, vec2(x, 0))`. Contract and arithmetic follow the reference's
shaderflow/audio/waveform.py:14-90:

    chunk_size = max(1, int(length*audio.samplerate/points))          (:65-67)
    offset     = audio.tell % chunk_size                              (:71-73)
    window     = history[-(chunk_size*points + offset + 1) : -(offset + 1)]      (:81-83)

i.e. chunks are aligned to multiples of `chunk_size` in stream time and the newest sample is excluded. The cut is
described to the device by (`tell`, `chunk_size`, `points`); the reductions run there (`sfx_waveform_rows`,
csrc/audio_kernels.hpp k_waveform_rows), one wavefront per chunk.
"""
from __future__ import annotations

import ctypes as C
import math
from collections.abc import Iterable
from enum import Enum
from typing import Sequence

import numpy as np
from attrs import define

from shaderflow_amd import _native as N
from shaderflow_amd.audio.module import BrokenAudio
from shaderflow_amd.module import ShaderModule
from shaderflow_amd.texture import ShaderTexture
from shaderflow_amd.variable import ShaderVariable, Uniform


class WaveformReducer(Enum):
    """How a chunk of samples becomes one point (device codes of include/shaderflow_hip.h SFX_REDUCER_*)"""
    Average = 0     # sqrt(mean(|x|))
    RMS = 1         # sqrt(sqrt(mean(x²))·√2)
    STD = 2         # sqrt(std(x))


@define(eq=False, slots=False)
class ShaderWaveform(ShaderModule):
    name: str = "iWaveform"
    audio: BrokenAudio = None
    length: float = 3
    """Seconds of history shown"""
    samplerate: float = 60
    """Points per second of history"""
    reducer: WaveformReducer = WaveformReducer.Average
    smooth: bool = True
    """Bilinear (True) or nearest (False) sampling of the row"""
    texture: ShaderTexture = None

    # geometry of the cut ------------------------------------------------------------------------------------

    @property
    def _points(self) -> int:
        return int(self.length*self.samplerate)

    @property
    def chunk_size(self) -> int:
        return max(1, int(self.length*self.audio.samplerate/self._points))

    @property
    def _offset(self) -> int:
        return self.audio.tell % self.chunk_size

    @property
    def _cutoff(self) -> int:
        return int(self.chunk_size*math.floor(self.audio.buffer_size/self.chunk_size))

    @property
    def length_samples(self) -> int:
        return int(max(1, self.length*self.scene.fps))

    # module -------------------------------------------------------------------------------------------------

    def build(self):
        self.texture = ShaderTexture(scene=self.scene, name=self.name, width=self._points, height=1,
                                     components=self.audio.channels, dtype=np.float32, mipmaps=False,
                                     filter=("linear" if self.smooth else "nearest"))
        self.texture.repeat(False)

    def rows(self, tells: Sequence[int]) -> np.ndarray:
        """(len(tells), points, channels) float32: the row each of those read positions produces"""
        pcm = getattr(self.audio, "native", None)
        if pcm is None:
            raise RuntimeError("The waveform's audio has no device-resident PCM: load a file into ShaderAudio first")
        tells = np.ascontiguousarray(tells, np.int64)
        out = np.zeros((len(tells), self._points, self.audio.channels), np.float32)
        N.check(N.lib().sfx_waveform_rows(pcm, N.as_ptr(tells, C.c_int64), len(tells), self.chunk_size, self._points,
                                          WaveformReducer(self.reducer).value, N.as_ptr(out, C.c_float)))
        return out

    def update(self):
        if self.texture.components != self.audio.channels:
            self.texture.components = self.audio.channels
        self.texture.write(self.rows([self.audio.tell])[0])

    def pipeline(self) -> Iterable[ShaderVariable]:
        yield Uniform("int", f"{self.name}Length", self.length_samples)
