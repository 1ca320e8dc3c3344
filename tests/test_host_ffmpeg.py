"""
The encoder command-line builder against argv produced by the reference's builder (tests/golden/ffmpeg_commands.json,
written by tests/golden/make_golden_ffmpeg.py from shaderflow/ffmpeg.py:753-1068). CPU only.
"""
import json
from pathlib import Path

import numpy as np
import pytest

from shaderflow_amd.ffmpeg import FFmpeg, Stage, pcm_dtype

GOLDEN = json.loads((Path(__file__).parent/"golden"/"ffmpeg_commands.json").read_text())


def build(fields, calls) -> FFmpeg:
    ffmpeg = FFmpeg()
    ffmpeg.executable = "ffmpeg"
    for name, value in fields.items():
        setattr(ffmpeg, name, value)
    for method, kwargs in calls:
        assert getattr(ffmpeg, method)(**kwargs) is ffmpeg               # fluent: every call returns the builder
    return ffmpeg


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: "+".join(m for m, _ in c["calls"])[:60])
def test_command_line_matches_the_reference_builder(case):
    assert list(build(case["fields"], case["calls"]).command) == case["argv"]


@pytest.mark.parametrize("case", GOLDEN["errors"], ids=lambda c: f"{len(c['calls'])}-calls")
def test_command_needs_input_and_output(case):
    with pytest.raises(ValueError) as info:
        build({}, case["calls"]).command
    assert str(info.value) == case["error"]


def test_defaults_and_recycling():
    ffmpeg = FFmpeg()
    assert ffmpeg.vcodec.kind == "h264" and ffmpeg.vcodec.crf == 20 and ffmpeg.acodec is None      # ffmpeg.py:811-815
    ffmpeg.pipe_input().scale(width=8, height=8).vflip().output("/tmp/x.mp4").aac()
    ffmpeg.clear(video_codec=False, audio_codec=False)                                             # exporting.py:91-92
    assert (ffmpeg.inputs, ffmpeg.filters, ffmpeg.outputs) == ([], [], []) and ffmpeg.vcodec.kind == "h264" and ffmpeg.acodec.kind == "aac"
    ffmpeg.clear()
    assert ffmpeg.vcodec is None and ffmpeg.acodec is None
    with pytest.raises(TypeError):
        ffmpeg.h264(quality=3)
    with pytest.raises(TypeError):
        ffmpeg.scale(width=3)
    with pytest.raises(AttributeError):
        ffmpeg.h264_nvenc()
    with pytest.raises(TypeError):
        ffmpeg.smartset("h264")
    ffmpeg.h264()
    ffmpeg.vcodec.crf = 17
    assert 17 in Stage.arguments(ffmpeg.vcodec, ffmpeg)


def test_device_vflip_drops_the_filter():
    a = FFmpeg().pipe_input().scale(width=64, height=36).vflip().output(path="/tmp/a.mp4")
    b = FFmpeg().pipe_input().scale(width=64, height=36).vflip(device=True).output(path="/tmp/a.mp4")
    a.executable = b.executable = "ffmpeg"
    assert "scale=64x36:flags=lanczos,vflip" in a.command and "scale=64x36:flags=lanczos" in b.command
    assert b.device_vflip and not a.device_vflip
    assert [x for x in a.command if "vflip" not in x] == [x for x in b.command if "scale" not in x]


def test_pcm_dtypes():
    assert pcm_dtype("pcm_f32le") == np.dtype("<f4") and pcm_dtype("pcm_s16be") == np.dtype(">i2")
    assert pcm_dtype("pcm_u8") == np.dtype("u1") and pcm_dtype("pcm_f64le") == np.dtype("<f8")


def test_audio_probes_answer_wav_files_natively(tmp_path):
    from shaderflow_amd.audio.reader import write_wav_f32
    samples = np.random.default_rng(0).uniform(-1, 1, (4410, 2)).astype(np.float32)
    path = write_wav_f32(tmp_path/"clip.wav", samples, 22050)
    assert FFmpeg.get_audio_samplerate(path) == 22050 and FFmpeg.get_audio_channels(path) == 2
    assert FFmpeg.get_audio_duration(path) == pytest.approx(0.2)
    assert np.array_equal(FFmpeg.get_audio_numpy(path), samples)
    assert FFmpeg.get_audio_samplerate(tmp_path/"missing.wav") is None and FFmpeg.get_video_resolution(tmp_path/"missing.mp4") is None


@pytest.mark.parametrize("claimed", [0, 0xFFFFFFFF, "double", "truncated"])
def test_probes_agree_with_the_decoder_on_streamed_and_truncated_wavs(tmp_path, claimed):
    """the duration a probe reports (it sizes the export: total frames) must be the duration the decoder delivers.
    A streamed / piped WAV leaves 0 or 0xFFFFFFFF in the data chunk's size; a truncated or over-promising one says more than is there."""
    import struct
    from shaderflow_amd.audio.reader import read_wav, write_wav_f32
    samples = np.random.default_rng(1).uniform(-1, 1, (1000, 2)).astype(np.float32)
    raw = bytearray(write_wav_f32(tmp_path/"clip.wav", samples, 8000).read_bytes())
    at = raw.index(b"data") + 4
    expected = samples
    if claimed == "double":
        raw[at:at + 4] = struct.pack("<I", 2*samples.nbytes)
    elif claimed == "truncated":
        raw, expected = raw[:-8*300], samples[:700]
    else:
        raw[at:at + 4] = struct.pack("<I", claimed)
    path = tmp_path/"odd.wav"
    path.write_bytes(bytes(raw))
    decoded, samplerate = read_wav(path)
    assert samplerate == 8000 and np.array_equal(decoded, expected)
    assert FFmpeg.get_audio_duration(path) == pytest.approx(len(expected)/8000)


FAKE_DECODER = """#!/usr/bin/env python3
import sys
src = sys.argv[sys.argv.index('-i') + 1]
assert 'pcm_f32le' in sys.argv and '-' in sys.argv
sys.stdout.buffer.write(open(src, 'rb').read()[4:])
"""
FAKE_PROBE = """#!/usr/bin/env python3
import sys
print('2' if 'stream=channels' in sys.argv else '32000')
"""


def test_non_wav_audio_goes_through_the_ffmpeg_binary(tmp_path, monkeypatch):
    """Containers other than RIFF/WAVE and FLAC (read natively) are decoded like the reference does (ffmpeg.py:1294-1301):
    pcm_f32le on a pipe, format from ffprobe; stand-in binaries here"""
    import os
    from shaderflow_amd.audio.reader import BrokenAudioReader, decode_audio
    samples = np.random.default_rng(1).uniform(-1, 1, (3000, 2)).astype(np.float32)
    (tmp_path/"song.ogg").write_bytes(b"OggS" + samples.tobytes())                 # not a real stream: the stand-in decoder strips the tag
    bindir = tmp_path/"bin"
    bindir.mkdir()
    (bindir/"ffmpeg").write_text(FAKE_DECODER)
    (bindir/"ffprobe").write_text(FAKE_PROBE)
    for tool in ("ffmpeg", "ffprobe"):
        (bindir/tool).chmod(0o755)
    monkeypatch.setenv("PATH", "/nonexistent")
    with pytest.raises(ValueError, match="no ffmpeg"):
        decode_audio(tmp_path/"song.ogg")
    monkeypatch.setenv("PATH", f"{bindir}{os.pathsep}/usr/bin{os.pathsep}/bin")
    decoded, samplerate = decode_audio(tmp_path/"song.ogg")
    assert samplerate == 32000 and np.array_equal(decoded, samples)
    reader = BrokenAudioReader(path=tmp_path/"song.ogg").load()
    assert reader.channels == 2 and reader.samplerate == 32000
