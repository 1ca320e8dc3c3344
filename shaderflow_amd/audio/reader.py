"""
PCM ingest for the export path.

The reference decodes audio by piping the file through an `ffmpeg` subprocess as pcm_f32le and slices that
stream with `BrokenAudioReader.stream` (shaderflow/ffmpeg.py:1240-1333). No ffmpeg binary is assumed here: RIFF/WAV
files (PCM 8/16/24/32-bit, IEEE float 32/64) and FLAC streams (the library's native decoder, csrc/flac.inc) are read into the
same float32 interleaved samples, and
`BrokenAudioReader` keeps the reference's chunk arithmetic — `target += chunk; length = (target - time)*Bps`
rounded to whole sample blocks, at least one block — because it defines which samples every frame sees
(frame 0 reads exactly ONE sample, frame 1 reads 734, then 735 per frame at 44.1 kHz / 60 fps).
"""
from __future__ import annotations

import struct
from collections.abc import Generator
from pathlib import Path
from typing import Optional

import numpy as np
from attrs import define


def read_wav(path: Path) -> tuple[np.ndarray, int]:
    """Returns (samples float32 (n, channels), samplerate)"""
    raw = Path(path).read_bytes()
    if raw[:4] != b"RIFF" or raw[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file (decode other containers to WAV first; no ffmpeg binary is used)")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        tag, size = raw[pos:pos + 4], struct.unpack("<I", raw[pos + 4:pos + 8])[0]
        body = raw[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:                 # WAVE_FORMAT_EXTENSIBLE: sub-format GUID
                fmt = (struct.unpack("<H", body[24:26])[0], *fmt[1:])
        elif tag == b"data":
            if size in (0, 0xFFFFFFFF):                              # a streamed / piped WAV: the writer never came back to patch the size
                body, size = raw[pos + 8:], len(raw) - pos - 8
            data = body                                              # (a truncated file: the slice ends with the file)
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    kind, channels, samplerate, _, _, bits = fmt
    if kind == 3 and bits == 32:
        pcm = np.frombuffer(data, "<f4").astype(np.float32)
    elif kind == 3 and bits == 64:
        pcm = np.frombuffer(data, "<f8").astype(np.float32)
    elif kind == 1 and bits == 16:
        pcm = (np.frombuffer(data, "<i2").astype(np.float32)/np.float32(32768.0))
    elif kind == 1 and bits == 32:
        pcm = (np.frombuffer(data, "<i4").astype(np.float64)/2147483648.0).astype(np.float32)
    elif kind == 1 and bits == 24:
        b = np.frombuffer(data[:len(data)//3*3], np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        pcm = (v.astype(np.float64)/8388608.0).astype(np.float32)
    elif kind == 1 and bits == 8:
        pcm = ((np.frombuffer(data, np.uint8).astype(np.float32) - 128.0)/128.0).astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAV format tag {kind} with {bits} bits")
    frames = pcm.size//channels
    return np.ascontiguousarray(pcm[:frames*channels].reshape(frames, channels)), int(samplerate)


def flac_info(path: Path) -> tuple[int, int, int, int]:
    """(samples per channel, channels, samplerate, bits) of a FLAC file, from its STREAMINFO block"""
    import ctypes as C

    from shaderflow_amd import _native as N
    with open(path, "rb") as file:
        raw = file.read(1 << 16)                                     # STREAMINFO is the first metadata block: no need for the audio
    samples, channels, samplerate, bits = C.c_int64(), C.c_int(), C.c_int(), C.c_int()
    N.check(N.lib().sfx_flac_info(raw, len(raw), C.byref(samples), C.byref(channels), C.byref(samplerate), C.byref(bits)))
    return samples.value, channels.value, samplerate.value, bits.value


def read_flac(path: Path) -> tuple[np.ndarray, int]:
    """Returns (samples float32 (n, channels), samplerate): the native decoder of the library (csrc/flac.inc)"""
    import ctypes as C

    from shaderflow_amd import _native as N
    raw = Path(path).read_bytes()
    samples, channels, samplerate, bits = C.c_int64(), C.c_int(), C.c_int(), C.c_int()
    N.check(N.lib().sfx_flac_info(raw, len(raw), C.byref(samples), C.byref(channels), C.byref(samplerate), C.byref(bits)))
    out = np.zeros((samples.value, channels.value), np.float32)
    written = C.c_int64()
    N.check(N.lib().sfx_flac_decode(raw, len(raw), N.as_ptr(out, C.c_float), out.size, C.byref(written)))
    return out[:written.value], int(samplerate.value)


def decode_audio(path: Path) -> tuple[np.ndarray, int]:
    """(samples float32 (n, channels), samplerate) of any audio file: RIFF/WAVE and FLAC natively, other containers through an
    `ffmpeg` binary as the reference does (pcm_f32le over a pipe, ffmpeg.py:1294-1301; samplerate and channels from ffprobe)"""
    raw = Path(path)
    with open(raw, "rb") as file:
        magic = file.read(12)
    if magic[:4] == b"RIFF" and magic[8:12] == b"WAVE":
        return read_wav(raw)
    if magic[:4] == b"fLaC":
        return read_flac(raw)
    import shutil
    import subprocess
    if not (shutil.which("ffmpeg") and shutil.which("ffprobe")):
        raise ValueError(f"{path}: neither RIFF/WAVE nor FLAC, and no ffmpeg/ffprobe binary to decode it with (convert it to WAV or FLAC first)")
    from shaderflow_amd.ffmpeg import FFmpeg
    samplerate, channels = FFmpeg.get_audio_samplerate(raw), FFmpeg.get_audio_channels(raw)
    command = FFmpeg().quiet().input(path=raw).pcm("pcm_f32le").no_video().output("-").command
    data = subprocess.run(command, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    pcm = np.frombuffer(data[:len(data)//(4*channels)*(4*channels)], "<f4")
    return np.ascontiguousarray(pcm.reshape(-1, channels)), int(samplerate)


def write_wav_f32(path: Path, samples: np.ndarray, samplerate: int) -> Path:
    """(n, channels) float32 → IEEE-float WAV (what the synthetic clips are stored as)"""
    samples = np.ascontiguousarray(samples, "<f4")
    channels = samples.shape[1]
    data = samples.tobytes()
    header = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(data), b"WAVE", b"fmt ", 16, 3, channels,
                         samplerate, samplerate*channels*4, channels*4, 32, b"data", len(data))
    Path(path).write_bytes(header + data)
    return Path(path)


@define(slots=False, eq=False)
class BrokenAudioReader:
    """Time-exact chunked reads of a decoded stream (ffmpeg.py:1240-1333)"""
    path: Optional[Path] = None
    samples: np.ndarray = None            # (n, channels) float32, the whole decoded file
    channels: int = None
    samplerate: int = None
    chunk: float = 0.1
    read: int = 0                         # bytes handed out so far
    bytes_per_sample: int = 4

    def load(self) -> "BrokenAudioReader":
        if self.samples is None:
            self.samples, self.samplerate = decode_audio(self.path)
        self.samples = np.ascontiguousarray(self.samples, np.float32)
        self.channels = self.samples.shape[1]
        return self

    @property
    def block_size(self) -> int:
        return (self.bytes_per_sample*self.channels)

    @property
    def bytes_per_second(self) -> int:
        return (self.block_size*self.samplerate)

    @property
    def time(self) -> float:
        return (self.read/self.bytes_per_second)

    def next_length(self, target: float) -> int:
        """Bytes the reference would request to reach `target` seconds (ffmpeg.py:1318-1321)"""
        length = (target - self.time)*self.bytes_per_second
        length = int(self.block_size*round(length/self.block_size))
        return max(length, self.block_size)

    @property
    def stream(self) -> Generator[np.ndarray, None, float]:
        self.load()
        self.read = 0
        total = self.samples.shape[0]*self.block_size
        target = 0
        while True:
            target += self.chunk
            length = min(self.next_length(target), total - self.read)
            if length <= 0:
                break
            first = self.read//self.block_size
            yield self.samples[first:first + length//self.block_size]
            self.read += length
        return self.time


def chunk_schedule(rdt: list[float], samplerate: int, channels: int, total_samples: int) -> np.ndarray:
    """`tell` after every frame of a whole export: the reader arithmetic above driven by the per-frame
    `rdt` the scene hands to `reader.chunk` (audio/module.py:450). int64 array, one entry per frame."""
    block = 4*channels
    bps = block*samplerate
    total = total_samples*block
    tell = np.zeros(len(rdt), np.int64)
    read, target, dry = 0, 0, False
    for k, chunk in enumerate(rdt):
        if not dry:
            target += chunk
            length = (target - (read/bps))*bps
            length = max(int(block*round(length/block)), block)
            length = min(length, total - read)
            if length <= 0:
                dry = True
            else:
                read += length
        tell[k] = read//block
    return tell
