"""
BrokenAudio / ShaderAudio (reference: shaderflow/audio/module.py:84-458), export path only.

`BrokenAudio` keeps the reference's observable state — `data` (channels, buffer_size) float32 history,
`tell`, `samplerate/channels/buffer_seconds`, `add_data`, `get_last_n_samples` (which EXCLUDES the newest
sample, :137-138) — but without rolling an 11 MB array every frame (:126): samples are appended to a linear
stream and `data` is materialised on demand. Soundcard capture/playback (:208-388) is realtime-only and absent.

`ShaderAudio` decodes the file once, keeps it resident in HBM (`sfx_audio_upload`), advances `tell` with the
reference's chunk arithmetic every frame and gets the loudness targets (`2*rms*sqrt(2)` and `np.std` over the
last 0.1 s, :457-458) from the device; the two ShaderDynamics (`<name>Volume` f=2 integrate, `<name>STD` f=10,
:413-421) are created in the same order so that `scene.modules` matches the reference.
"""
from __future__ import annotations

import ctypes as C
import math
from collections.abc import Generator
from enum import Enum
from pathlib import Path
from typing import Optional

import numpy as np
from attrs import define, field

from shaderflow_amd import _native as N
from shaderflow_amd.audio.reader import BrokenAudioReader
from shaderflow_amd.dynamics import ShaderDynamics
from shaderflow_amd.module import ShaderModule


def root_mean_square(data) -> float:
    return np.sqrt(np.mean(np.square(data)))


class AudioMode(Enum):
    Realtime = "realtime"
    File = "file"


@define(slots=False, eq=False)
class BrokenAudio:
    mode: AudioMode = field(default=AudioMode.Realtime, converter=AudioMode)
    dtype: np.dtype = np.float32
    tell: int = 0
    """Samples appended so far"""

    _samplerate: float = 44100
    _channels: int = 2
    _buffer_seconds: float = 30.0
    _stream: np.ndarray = None            # (channels, capacity) linear history
    _length: int = 0

    def __attrs_post_init__(self):
        self.create_buffer()

    @property
    def buffer_size(self) -> int:
        return int(self.samplerate*self.buffer_seconds)

    @property
    def shape(self) -> tuple[int, int]:
        return (self.channels, self.buffer_size)

    def create_buffer(self) -> None:
        self._stream = np.zeros((self.channels, 0), dtype=self.dtype)
        self._length = 0

    @property
    def data(self) -> np.ndarray:
        """The reference's ring: the last `buffer_size` samples, zeros before the stream started (:110-111)"""
        out = np.zeros(self.shape, dtype=self.dtype)
        n = min(self._length, self.buffer_size)
        if n:
            out[:, -n:] = self._stream[:, self._length - n:self._length]
        return out

    def add_data(self, data: np.ndarray) -> Optional[np.ndarray]:
        data = np.array(data, dtype=self.dtype)
        length = data.shape[1]
        if self._length + length > self._stream.shape[1]:
            grown = np.zeros((self.channels, max(2*self._stream.shape[1], self._length + length, 1 << 16)), self.dtype)
            grown[:, :self._length] = self._stream[:, :self._length]
            self._stream = grown
        self._stream[:, self._length:self._length + length] = data
        self._length += length
        self.tell += length
        return data

    def _window(self, first: int, last: int) -> np.ndarray:
        """stream[first:last] with zeros outside [0, length)"""
        out = np.zeros((self.channels, max(0, last - first)), dtype=self.dtype)
        lo, hi = max(first, 0), min(last, self._length)
        if hi > lo:
            out[:, lo - first:hi - first] = self._stream[:, lo:hi]
        return out

    def get_last_n_samples(self, n: int, *, offset: int = 0) -> np.ndarray:
        # data[:, -(n+offset+1) : -(offset+1)] of the ring (audio/module.py:137-138)
        end = self._length - int(offset) - 1
        return self._window(self._length - int(n + offset) - 1, end)

    def get_last_n_seconds(self, n: float) -> np.ndarray:
        return self.get_last_n_samples(n*self.samplerate)

    @property
    def samplerate(self) -> float:
        return (self._samplerate or 44100)

    @samplerate.setter
    def samplerate(self, value: float):
        self._samplerate = value
        self.create_buffer()

    @property
    def channels(self) -> int:
        return self._channels or 2

    @channels.setter
    def channels(self, value: int):
        self._channels = value
        self.create_buffer()

    @property
    def buffer_seconds(self) -> float:
        return self._buffer_seconds

    @buffer_seconds.setter
    def buffer_seconds(self, value: float):
        self._buffer_seconds = value
        self.create_buffer()

    @property
    def stereo(self) -> bool:
        return (self.channels == 2)

    @property
    def mono(self) -> bool:
        return (self.channels == 1)


@define(slots=False, eq=False)
class ShaderAudio(BrokenAudio, ShaderModule):
    volume: ShaderDynamics = None
    std: ShaderDynamics = None
    final: bool = True

    _file: Path = None
    _file_reader: BrokenAudioReader = None
    _file_stream: Generator = None
    native: Optional[N.Handle] = None
    """Device-resident PCM of the whole file (sfx_audio_upload)"""

    def __attrs_post_init__(self):
        BrokenAudio.__attrs_post_init__(self)
        ShaderModule.__attrs_post_init__(self)
        self.volume = ShaderDynamics(
            scene=self.scene, name=f"{self.name}Volume",
            frequency=2, zeta=1, response=0, value=0, integrate=True,
        )
        self.std = ShaderDynamics(
            scene=self.scene, name=f"{self.name}STD",
            frequency=10, zeta=1, response=0, value=0,
        )

    # file -------------------------------------------------------------------------------------------------

    @property
    def file(self) -> Optional[Path]:
        return self._file

    @file.setter
    def file(self, value):
        if value is None:
            return
        self.load(path=Path(value))

    def load(self, path: Optional[Path] = None, samples: Optional[np.ndarray] = None, samplerate: Optional[int] = None):
        """Decode (or take `samples` (n, channels) float32), upload to HBM, arm the chunk reader"""
        if path is not None and not Path(path).exists():
            self.log_warn(f"Audio File doesn't exist ({path})")
            return self
        self._file = Path(path) if path is not None else None
        reader = BrokenAudioReader(path=self._file, samples=samples, samplerate=samplerate).load()
        self._samplerate, self._channels = reader.samplerate, reader.channels
        self.create_buffer()
        self.tell = 0
        self._file_reader = reader
        self._file_stream = reader.stream
        self.mode = AudioMode.File
        self._upload(reader.samples)
        return self

    def _upload(self, samples: np.ndarray) -> None:
        self.release()
        handle = N.Handle()
        flat = np.ascontiguousarray(samples, np.float32)
        N.check(N.lib().sfx_audio_upload(self.scene.context.handle, N.as_ptr(flat, C.c_float), flat.shape[0],
                                         flat.shape[1], int(self.samplerate), C.byref(handle)))
        self.native = handle

    def release(self) -> None:
        if self.native is not None and self.native.value:
            N.lib().sfx_audio_destroy(self.native)
        self.native = None

    def destroy(self) -> None:
        self.release()

    @property
    def duration(self) -> float:
        if self._file_reader is None:
            return 0.0
        return self._file_reader.samples.shape[0]/self.samplerate

    def setup(self):
        if self._file_reader is not None:                    # `self.file = self.file` re-arms the stream (:435)
            self._file_stream = self._file_reader.stream
            self.create_buffer()
            self.tell = 0

    def ffhook(self, ffmpeg) -> None:
        if (self.file is not None) and self.file.exists():
            ffmpeg.input(path=self.file)
            ffmpeg.shortest = True

    # frame --------------------------------------------------------------------------------------------------

    def loudness_targets(self, tells) -> np.ndarray:
        """[(volume target, std target)] for each `tell` (audio/module.py:457-458), computed on the device"""
        tells = np.ascontiguousarray(tells, np.int64)
        out = np.zeros((len(tells), 2), np.float32)
        N.check(N.lib().sfx_volume_std(self.native, N.as_ptr(tells, C.c_int64), len(tells),
                                       int(0.1*self.samplerate), N.as_ptr(out, C.c_float)))
        return out

    def update(self):
        try:
            if self._file_stream:
                self._file_reader.chunk = self.scene.rdt
                data = next(self._file_stream).T
                self.add_data(data)
        except StopIteration:
            pass

        if self.native is not None:
            volume, std = self.loudness_targets([self.tell])[0]
            self.volume.target = np.float32(volume)
            self.std.target = np.float32(std)
        else:
            # no file loaded: the history is all zeros (audio/module.py:110-111), both targets are exactly 0
            self.volume.target = np.float32(0.0)
            self.std.target = np.float32(0.0)
