"""
FFmpeg: fluent builder of the encoder command line an export feeds its frames to.

Host mirror of the builder half of the reference's shaderflow/ffmpeg.py:753-1075 (what `ExportingHelper` and
`ShaderModule.ffhook` call: exporting.py:91-120, audio/module.py:442-445): the same method names, option names,
defaults and argument order on the command line, so user code such as `scene.ffmpeg.h264(crf=18)` or
`ffmpeg.input(path=...)` keeps working and produces the same argv (pinned by tests/golden/ffmpeg_commands.json,
generated from the reference builder).

Here every stage (input, output, codec, filter) is one `Stage` record: a kind, its option values, and a row of the
`STAGES` table that says how the options become arguments. An argument group is dropped when one of its values
is None or "" (the reference's `every()` helper, ffmpeg.py:36-39). Hardware encoders of other vendors (nvenc/qsv)
are intentionally not restated; AMD's AMF names are accepted.

Device-side extension: `FFmpeg.vflip(device=True)` records that the frames arrive top-down
already (the read-out kernels flip rows while writing), so no `vflip` filter is emitted.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from enum import Enum
from pathlib import Path
from typing import Any, Callable, Iterable, Optional

from shaderflow_amd.module import logger

REQUIRED = object()


def _plain(value: Any) -> Any:
    return value.value if isinstance(value, Enum) else value


def _group(*items: Any) -> tuple:
    """The items, or nothing when one of them is missing"""
    items = tuple(_plain(item) for item in items)
    return () if any(item is None or item == "" for item in items) else items


class Stage:
    """One configured element of the command line"""
    __slots__ = ("kind", "role", "options")

    def __init__(self, kind: str, **options: Any):
        role, defaults, _ = STAGES[kind]
        unknown = set(options) - set(defaults)
        if unknown:
            raise TypeError(f"{kind}() got unexpected option(s) {sorted(unknown)}; known: {sorted(defaults)}")
        missing = [name for name, default in defaults.items() if default is REQUIRED and name not in options]
        if missing:
            raise TypeError(f"{kind}() missing required option(s) {missing}")
        self.kind, self.role = kind, role
        self.options = {**defaults, **options}

    def __getattr__(self, name: str) -> Any:
        try:
            return self.options[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name: str, value: Any) -> None:
        if name in Stage.__slots__:
            object.__setattr__(self, name, value)
        elif name in self.options:
            self.options[name] = value
        else:
            raise AttributeError(f"{self.kind} has no option {name!r}")

    def arguments(self, ffmpeg: "FFmpeg") -> tuple:
        return tuple(STAGES[self.kind][2](self.options, ffmpeg))

    def __str__(self) -> str:                       # filters are joined with "," by str()
        return ",".join(map(str, self.arguments(None)))

    def __repr__(self) -> str:
        return f"Stage({self.kind!r}, {self.options})"


def _lavfi_silence(o: dict, ff: "FFmpeg") -> tuple:
    return ("-f", "lavfi") + (("-t", ff.time) if ff.time else ()) + ("-i", f"anullsrc=channel_layout=stereo:sample_rate={o['samplerate']}")


# kind → (role, {option: default}, emitter(options, ffmpeg) → arguments)
STAGES: dict[str, tuple[str, dict[str, Any], Callable[[dict, Any], Iterable]]] = {
    # inputs (ffmpeg.py:53-86)
    "input": ("input", {"path": REQUIRED}, lambda o, ff: ("-i", o["path"])),
    "pipe_input": ("input", {"format": "rawvideo", "pixel_format": "rgb24", "width": 1920, "height": 1080, "framerate": 60.0},
                   lambda o, ff: ("-f", _plain(o["format"]), "-s", f"{o['width']}x{o['height']}", "-pix_fmt", _plain(o["pixel_format"]),
                                  "-r", o["framerate"], "-i", "-")),
    # outputs (ffmpeg.py:97-136)
    "output": ("output", {"path": REQUIRED, "pixel_format": "yuv420p", "overwrite": True},
               lambda o, ff: _group("-pix_fmt", o["pixel_format"]) + (o["path"], "-y" if o["overwrite"] else "")),
    "pipe_output": ("output", {"format": "mpegts", "pixel_format": None},
                    lambda o, ff: _group("-f", o["format"]) + _group("-pix_fmt", o["pixel_format"]) + ("pipe:1",)),
    # video codecs (ffmpeg.py:149-207, 296-323, 414-455, 534-549)
    "h264": ("vcodec", {"preset": "slow", "tune": None, "profile": None, "crf": 20, "bitrate": None, "x264params": ()},
             lambda o, ff: ("-c:v", "libx264", "-movflags", "+faststart") + _group("-profile", o["profile"]) + _group("-preset", o["preset"])
             + _group("-tune", o["tune"]) + _group("-b:v", o["bitrate"]) + _group("-crf", o["crf"])
             + _group("-x264opts", ":".join(o["x264params"] or ()))),
    "h265": ("vcodec", {"crf": 25, "bitrate": None, "preset": "slow"},
             lambda o, ff: ("-c:v", "libx265") + _group("-preset", o["preset"]) + _group("-crf", o["crf"]) + _group("-b:v", o["bitrate"])),
    "av1_svt": ("vcodec", {"crf": 25, "preset": 3},
                lambda o, ff: ("-c:v", "libsvtav1", "-crf", o["crf"], "-preset", o["preset"], "-svtav1-params", "tune=0")),
    "av1_rav1e": ("vcodec", {"qp": 80, "speed": 4, "tile_rows": 4, "tile_columns": 4},
                  lambda o, ff: ("-c:v", "librav1e", "-qp", o["qp"], "-speed", o["speed"], "-tile-rows", o["tile_rows"],
                                 "-tile-columns", o["tile_columns"])),
    "h264_amf": ("vcodec", {"quality": "quality", "bitrate": None}, lambda o, ff: ("-c:v", "h264_amf") + _group("-quality", o["quality"]) + _group("-b:v", o["bitrate"])),
    "h265_amf": ("vcodec", {"quality": "quality", "bitrate": None}, lambda o, ff: ("-c:v", "hevc_amf") + _group("-quality", o["quality"]) + _group("-b:v", o["bitrate"])),
    "av1_amf": ("vcodec", {"quality": "quality", "bitrate": None}, lambda o, ff: ("-c:v", "av1_amf") + _group("-quality", o["quality"]) + _group("-b:v", o["bitrate"])),
    "rawvideo": ("vcodec", {}, lambda o, ff: ("-c:v", "rawvideo")),
    "no_video": ("vcodec", {}, lambda o, ff: ("-c:v", "null")),
    "copy_video": ("vcodec", {}, lambda o, ff: ("-c:v", "copy")),
    # audio codecs (ffmpeg.py:567-697)
    "aac": ("acodec", {"bitrate": 192}, lambda o, ff: ("-c:a", "aac", "-b:a", f"{o['bitrate']}k")),
    "mp3": ("acodec", {"bitrate": 192, "qscale": 2}, lambda o, ff: ("-c:a", "libmp3lame", "-b:a", f"{o['bitrate']}k") + _group("-qscale:a", o["qscale"])),
    "opus": ("acodec", {"bitrate": 192}, lambda o, ff: ("-c:a", "libopus", "-b:a", f"{o['bitrate']}k")),
    "flac": ("acodec", {}, lambda o, ff: ("-c:a", "flac")),
    "copy_audio": ("acodec", {}, lambda o, ff: ("-c:a", "copy")),
    "no_audio": ("acodec", {}, lambda o, ff: ("-an",)),
    "empty_audio": ("acodec", {"samplerate": 44100}, _lavfi_silence),
    "pcm": ("acodec", {"format": "pcm_f32le"}, lambda o, ff: ("-c:a", _plain(o["format"]), "-f", str(_plain(o["format"])).removeprefix("pcm_"))),
    # filters (ffmpeg.py:711-741)
    "scale": ("filter", {"width": REQUIRED, "height": REQUIRED, "resample": "lanczos"},
              lambda o, ff: (f"scale={o['width']}x{o['height']}:flags={_plain(o['resample'])}",)),
    "vflip": ("filter", {}, lambda o, ff: ("vflip",)),
    "filter": ("filter", {"content": REQUIRED}, lambda o, ff: (o["content"],)),
}


def pcm_dtype(format: str):
    """numpy dtype of a raw `pcm_*` format name (ffmpeg.py:638-673): pcm_f32le → '<f4', pcm_s16be → '>i2'"""
    import numpy as np
    name = str(_plain(format))
    size = int("".join(c for c in name if c.isdigit()))//8
    kind = {"s": "i", "u": "u", "f": "f"}[name[4]]                    # (the reference maps 's' straight to numpy, which rejects it)
    return np.dtype(f"{'<' if 'le' in name else '>'}{kind}{size}")


class FFmpeg:
    """Fluent command-line builder; every configuring method returns self"""

    def __init__(self, *, hide_banner: bool = True, shortest: bool = False, stream_loop: int = 0, time: Optional[float] = None,
                 vsync: str = "cfr", loglevel: str = "error", hwaccel: Optional[str] = None):
        self.hide_banner, self.shortest, self.stream_loop, self.time = hide_banner, shortest, stream_loop, time
        self.vsync, self.loglevel, self.hwaccel = vsync, loglevel, hwaccel
        self.inputs: list[Stage] = []
        self.filters: list[Stage] = []
        self.outputs: list[Stage] = []
        self.vcodec: Optional[Stage] = Stage("h264")
        self.acodec: Optional[Stage] = None
        self.device_vflip: bool = False

    # every stage kind is a method: ffmpeg.h264(crf=18), ffmpeg.scale(width=…, height=…), ffmpeg.input(path=…)
    def __getattr__(self, kind: str):
        if kind.startswith("_") or kind not in STAGES:
            raise AttributeError(kind)
        return lambda *args, **options: self.smartset(_stage_from_call(kind, args, options))

    def vflip(self, device: bool = False) -> "FFmpeg":
        """`device=True`: the frames are flipped on the GPU while they are written, no filter is needed"""
        if device:
            self.device_vflip = True
            return self
        return self.smartset(Stage("vflip"))

    def smartset(self, stage: Stage) -> "FFmpeg":
        if not isinstance(stage, Stage):
            raise TypeError(f"Unsupported type: {type(stage)}")
        if stage.role in ("vcodec", "acodec"):
            setattr(self, stage.role, stage)
        else:
            getattr(self, stage.role + "s").append(stage)
        return self

    def quiet(self) -> "FFmpeg":
        self.hide_banner, self.loglevel = True, "error"
        return self

    # recycling (ffmpeg.py:827-859) -----------------------------------------------------------------------

    def clear_inputs(self) -> "FFmpeg":
        self.inputs = []
        return self

    def clear_filters(self) -> "FFmpeg":
        self.filters, self.device_vflip = [], False
        return self

    def clear_outputs(self) -> "FFmpeg":
        self.outputs = []
        return self

    def clear_video_codec(self) -> "FFmpeg":
        self.vcodec = None
        return self

    def clear_audio_codec(self) -> "FFmpeg":
        self.acodec = None
        return self

    def clear(self, inputs: bool = True, filters: bool = True, outputs: bool = True, video_codec: bool = True, audio_codec: bool = True) -> "FFmpeg":
        for wanted, method in ((inputs, self.clear_inputs), (filters, self.clear_filters), (outputs, self.clear_outputs),
                               (video_codec, self.clear_video_codec), (audio_codec, self.clear_audio_codec)):
            if wanted:
                method()
        return self

    # command line (ffmpeg.py:1027-1068) ------------------------------------------------------------------

    executable: Optional[str] = None

    @property
    def command(self) -> tuple[str, ...]:
        if not self.inputs:
            raise ValueError("At least one input is required for FFmpeg")
        if not self.outputs:
            raise ValueError("At least one output is required for FFmpeg")
        argv: list[Any] = [self.executable or shutil.which("ffmpeg")]
        if self.hide_banner:
            argv.append("-hide_banner")
        argv += ["-loglevel", _plain(self.loglevel)]
        if self.hwaccel is not None:
            argv += ["-hwaccel", _plain(self.hwaccel)]
        if self.stream_loop > 0:
            argv += ["-stream_loop", self.stream_loop]
        for stage in self.inputs:
            argv += stage.arguments(self)
        if self.time is not None:
            argv += ["-t", self.time]
        if self.shortest:
            argv.append("-shortest")
        for output in self.outputs:
            for codec in (self.acodec, self.vcodec):
                if codec is not None:
                    argv += codec.arguments(self)
            if self.filters:
                argv += ["-vf", ",".join(map(str, self.filters))]
            argv += output.arguments(self)
        return tuple(map(str, argv))

    def run(self, **options) -> subprocess.CompletedProcess:
        return subprocess.run(self.command, **options)

    def popen(self, **options) -> subprocess.Popen:
        logger.info(f"Call {self.command}")
        return subprocess.Popen(self.command, **options)

    @staticmethod
    def available() -> bool:
        return shutil.which("ffmpeg") is not None

    # media probes (ffmpeg.py:1104-1235) ----------------------------------------------------------------------
    # RIFF/WAVE and FLAC files are answered by the native readers; anything else needs the ffprobe binary.
    # A path that does not exist gives None, like the reference.

    @staticmethod
    def _wav(path: Path):
        """(samples, samplerate) of the containers read natively — RIFF/WAVE and FLAC, by their magic bytes — else None"""
        from shaderflow_amd.audio.reader import read_flac, read_wav
        with open(path, "rb") as file:
            magic = file.read(12)
        if magic[:4] == b"RIFF" and magic[8:12] == b"WAVE":
            return read_wav(path)
        if magic[:4] == b"fLaC":
            return read_flac(path)
        return None

    @staticmethod
    def _header(path: Path):
        """(samples per channel, channels, samplerate) of a RIFF/WAVE or FLAC file from its HEADER — the probes below answer from it
        without decoding (a FLAC file used to be decoded by each of them); None for other containers"""
        import struct
        from shaderflow_amd.audio.reader import flac_info
        with open(path, "rb") as file:
            head = file.read(12)
            if head[:4] == b"fLaC":
                samples, channels, samplerate, _ = flac_info(path)      # STREAMINFO only (sfx_flac_info)
                return samples, channels, samplerate
            if not (head[:4] == b"RIFF" and head[8:12] == b"WAVE"):
                return None
            channels = samplerate = block = None
            while len(chunk := file.read(8)) == 8:
                tag, size = chunk[:4], struct.unpack("<I", chunk[4:])[0]
                if tag == b"fmt ":
                    body = file.read(size + (size & 1))
                    _, channels, samplerate, _, block, _ = struct.unpack("<HHIIHH", body[:16])
                elif tag == b"data":
                    if channels is None or not block:
                        return None
                    # what a decoder will find, not what the header promises: a streamed WAV says 0 or 0xFFFFFFFF, a truncated one too much
                    rest = os.fstat(file.fileno()).st_size - file.tell()
                    size = rest if size in (0, 0xFFFFFFFF) else min(size, rest)
                    return size//block, channels, samplerate
                else:
                    file.seek(size + (size & 1), 1)
        return None

    @staticmethod
    def _probe(path: Path, stream: str, entry: str) -> str:
        if shutil.which("ffprobe") is None:
            raise RuntimeError(f"{path}: probing this container needs the ffprobe binary (RIFF/WAVE and FLAC files do not)")
        return subprocess.check_output(["ffprobe", "-hide_banner", "-loglevel", "error", "-select_streams", stream,
                                        "-show_entries", entry, "-of", "default=noprint_wrappers=1:nokey=1", str(path)], text=True).strip()

    @staticmethod
    def get_audio_samplerate(path: Path, *, stream: int = 0, echo: bool = True) -> Optional[int]:
        if not (path := Path(path)).exists():
            return None
        header = FFmpeg._header(path)
        return header[2] if header else int(FFmpeg._probe(path, f"a:{stream}", "stream=sample_rate"))

    @staticmethod
    def get_audio_channels(path: Path, *, stream: int = 0, echo: bool = True) -> Optional[int]:
        if not (path := Path(path)).exists():
            return None
        header = FFmpeg._header(path)
        return header[1] if header else int(FFmpeg._probe(path, f"a:{stream}", "stream=channels"))

    @staticmethod
    def get_audio_duration(path: Path, *, echo: bool = True) -> Optional[float]:
        if not (path := Path(path)).exists():
            return None
        header = FFmpeg._header(path)
        return header[0]/header[2] if header else float(FFmpeg._probe(path, "a:0", "format=duration"))

    @staticmethod
    def get_audio_numpy(path: Path, *, echo: bool = True):
        """(samples, channels) float32 of the whole file"""
        if not (path := Path(path)).exists():
            return None
        wav = FFmpeg._wav(path)
        if wav:
            return wav[0]
        raise RuntimeError(f"{path}: decoding this container needs an ffmpeg binary; convert it to WAV or FLAC")

    @staticmethod
    def get_video_resolution(path: Path, *, echo: bool = True) -> Optional[tuple[int, int]]:
        if not (path := Path(path)).exists():
            return None
        width, height = FFmpeg._probe(path, "v:0", "stream=width,height").split()[:2]
        return int(width), int(height)

    @staticmethod
    def get_video_framerate(path: Path, *, precise: bool = False, echo: bool = True) -> Optional[float]:
        if not (path := Path(path)).exists():
            return None
        num, _, den = FFmpeg._probe(path, "v:0", "stream=r_frame_rate").partition("/")
        return float(num)/float(den or 1)

    @staticmethod
    def get_video_duration(path: Path, *, echo: bool = True) -> Optional[float]:
        if not (path := Path(path)).exists():
            return None
        return float(FFmpeg._probe(path, "v:0", "format=duration"))


def _stage_from_call(kind: str, args: tuple, options: dict) -> Stage:
    """Positional sugar the reference allows: input(path), output(path), filter(content), pcm(format)"""
    if args:
        names = [name for name in STAGES[kind][1]]
        if len(args) > 1 or not names:
            raise TypeError(f"{kind}() takes at most one positional argument")
        options = {names[0]: args[0], **options}
    if "path" in options and options["path"] is not None:
        options["path"] = Path(options["path"])
    return Stage(kind, **options)
