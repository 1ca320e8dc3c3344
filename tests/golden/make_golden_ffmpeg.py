#!/usr/bin/env python3
"""
Golden encoder command lines: drives the REFERENCE's FFmpeg builder (shaderflow/ffmpeg.py:753-1068, read-only at
/root/reference) through a list of call chains and stores each chain next to the argv it produced in
ffmpeg_commands.json. Runs only in the build container; the test (tests/test_host_ffmpeg.py) replays the same
chains on shaderflow_amd.ffmpeg.FFmpeg. argv[0] (the resolved binary) is stored as "ffmpeg".
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden  # noqa: E402  (installs the import shims, imports the reference)

RefFFmpeg = make_golden.RealFFmpeg

# a chain = [fields set on the object, [(method, kwargs), …]]
CHAINS = [
    # what ExportingHelper builds for a 1080p60 10 s export (exporting.py:91-116) + the audio hook (audio/module.py:442-445)
    [{"time": 10.0}, [("clear", {"video_codec": False, "audio_codec": False}),
                      ("pipe_input", {"pixel_format": "rgb24", "width": 1920, "height": 1080, "framerate": 60.0}),
                      ("scale", {"width": 1920, "height": 1080}), ("vflip", {}), ("output", {"path": "/tmp/video.mp4"})]],
    [{"time": 60.0, "shortest": True}, [("pipe_input", {"pixel_format": "rgb24", "width": 3840, "height": 2160, "framerate": 60.0}),
                                         ("scale", {"width": 3840, "height": 2160}), ("vflip", {}), ("input", {"path": "/tmp/song.wav"}),
                                         ("output", {"path": "/tmp/video.mp4"})]],
    [{}, [("pipe_input", {"width": 1280, "height": 720, "framerate": 30}), ("h264", {"crf": 18, "preset": "fast", "tune": "animation", "profile": "high"}),
          ("aac", {"bitrate": 256}), ("output", {"path": "/tmp/a.mkv", "pixel_format": "yuv444p", "overwrite": False})]],
    [{}, [("pipe_input", {"pixel_format": "rgba", "width": 64, "height": 64, "framerate": 24.0}), ("h264", {"bitrate": 8000, "x264params": ["keyint=30", "bframes=2"]}),
          ("output", {"path": "/tmp/b.mp4"})]],
    [{}, [("pipe_input", {}), ("h265", {}), ("opus", {}), ("output", {"path": "/tmp/c.mkv"})]],
    [{}, [("pipe_input", {}), ("h265", {"crf": 30, "bitrate": 4000, "preset": "medium"}), ("mp3", {"qscale": 4}), ("output", {"path": "/tmp/c.mkv"})]],
    [{}, [("pipe_input", {}), ("av1_svt", {"crf": 30, "preset": 6}), ("flac", {}), ("output", {"path": "/tmp/d.webm"})]],
    [{}, [("pipe_input", {}), ("av1_rav1e", {}), ("no_audio", {}), ("output", {"path": "/tmp/e.mkv"})]],
    [{"time": 3.5}, [("pipe_input", {}), ("rawvideo", {}), ("empty_audio", {}), ("pipe_output", {"format": "rawvideo", "pixel_format": "rgb24"})]],
    [{"loglevel": "info", "hide_banner": False, "stream_loop": 2, "hwaccel": "auto"},
     [("input", {"path": "/tmp/in.mp4"}), ("copy_video", {}), ("copy_audio", {}), ("filter", {"content": "eq=gamma=1.2"}), ("pipe_output", {})]],
    [{}, [("input", {"path": "/tmp/in.flac"}), ("clear_video_codec", {}), ("pcm", {"format": "pcm_s16le"}), ("pipe_output", {"format": "null"})]],
    [{}, [("pipe_input", {}), ("scale", {"width": 640, "height": 360, "resample": "bicubic"}), ("filter", {"content": "hue=s=0"}), ("vflip", {}),
          ("no_video", {}), ("output", {"path": "/tmp/f.mp4"}), ("output", {"path": "/tmp/g.mp4"})]],
]


def main() -> None:
    cases = []
    for fields, calls in CHAINS:
        ffmpeg = RefFFmpeg()
        for name, value in fields.items():
            setattr(ffmpeg, name, value)
        for method, kwargs in calls:
            if method == "pcm":                      # the reference wants its enum here; the value is what is stored
                kwargs = {"format": make_golden.ref_ffmpeg.FFmpegPCM(kwargs["format"])}
            getattr(ffmpeg, method)(**kwargs)
        argv = list(ffmpeg.command)
        argv[0] = "ffmpeg"
        cases.append({"fields": fields, "calls": [[m, k] for m, k in calls], "argv": argv})
    # error behaviour
    errors = []
    for calls in ([], [("pipe_input", {})]):
        ffmpeg = RefFFmpeg()
        for method, kwargs in calls:
            getattr(ffmpeg, method)(**kwargs)
        try:
            ffmpeg.command
            errors.append({"calls": [[m, k] for m, k in calls], "error": None})
        except ValueError as error:
            errors.append({"calls": [[m, k] for m, k in calls], "error": str(error)})
    out = Path(__file__).with_name("ffmpeg_commands.json")
    out.write_text(json.dumps({"cases": cases, "errors": errors}, indent=1))
    print(f"{out.name}: {len(cases)} command lines, {len(errors)} error cases")


if __name__ == "__main__":
    main()
