"""
Native FLAC ingest (csrc/flac.inc through shaderflow_amd.audio.reader) on the CPU: streams written by tests/flac_encoder.py — every
subframe type, both Rice parameter widths, escape partitions, wasted bits, all channel assignments, odd block sizes, 8/16/24-bit —
must decode to exactly the samples that went in (the same samples as a WAV round trip), and corrupted streams must fail loudly.
"""
import struct

import numpy as np
import pytest

from shaderflow_amd.audio.reader import decode_audio, flac_info, read_flac, read_wav
from tests.flac_encoder import encode


def music(n: int, channels: int, bits: int, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    t = np.arange(n)/44100.0
    peak = (1 << (bits - 1)) - 1
    left = 0.6*np.sin(2*np.pi*440*t) + 0.2*np.sin(2*np.pi*1234.5*t + 1) + 0.01*rng.standard_normal(n)
    right = 0.5*np.sin(2*np.pi*330*t + 0.5) + 0.25*np.sin(2*np.pi*(200 + 300*t)*t) + 0.01*rng.standard_normal(n)
    waves = np.stack([left, right, 0.5*(left - right)][:channels], axis=1)
    return np.clip(np.rint(waves*peak*0.9), -peak - 1, peak).astype(np.int64)


def as_float(pcm: np.ndarray, bits: int) -> np.ndarray:
    return (pcm.astype(np.float64)/float(1 << (bits - 1))).astype(np.float32)


@pytest.mark.parametrize("bits", [8, 16, 24])
@pytest.mark.parametrize("assignment", [None, 8, 9, 10])
def test_fixed_predictors_and_channel_assignments_round_trip(tmp_path, bits, assignment):
    pcm = music(5000, 2, bits, seed=bits)
    orders = [0, 1, 2, 3, 4]
    plan = lambda frame, channels: dict(assignment=assignment, subframes=[dict(kind="fixed", order=orders[(frame + c) % 5], partition_order=(frame + c) % 4,
                                                                              wide=bool((frame + c) % 2)) for c in range(channels)])
    path = tmp_path/"clip.flac"
    path.write_bytes(encode(pcm, 44100, bits, blocksize=1152, plan=plan))
    assert flac_info(path) == (5000, 2, 44100, bits)
    samples, samplerate = read_flac(path)
    assert samplerate == 44100 and samples.shape == (5000, 2) and samples.dtype == np.float32
    assert np.array_equal(samples, as_float(pcm, bits))
    again, _ = decode_audio(path)                                      # the reader's entry point picks the decoder by the magic bytes
    assert np.array_equal(again, samples)


def test_lpc_constant_verbatim_escape_and_wasted_bits(tmp_path):
    pcm = music(4096*3 + 777, 2, 16, seed=3)
    pcm[4096:8192, 1] = -1234                                          # a constant block
    pcm[8192:12288, 0] = (pcm[8192:12288, 0] >> 3) << 3                # three wasted bits
    lpc8 = ([1820, -1100, 420, -90, 30, -12, 5, -2], 12, 10)           # order 8, 12-bit coefficients, shift 10
    lpc2 = ([2047, -1023], 12, 10)

    def plan(frame, channels):
        if frame == 0:
            return dict(subframes=[dict(kind="lpc", order=8, lpc=lpc8, partition_order=3), dict(kind="lpc", order=2, lpc=lpc2, partition_order=0, wide=True)])
        if frame == 1:
            return dict(subframes=[dict(kind="verbatim"), dict(kind="constant")])
        if frame == 2:
            return dict(subframes=[dict(kind="fixed", order=2, wasted=3, partition_order=2, escape_partitions=(1, 3)), dict(kind="lpc", order=32, lpc=([3]*32, 5, 7), partition_order=1)])
        return dict(assignment=10, subframes=[dict(kind="fixed", order=1), dict(kind="fixed", order=4, escape_partitions=(0,))])

    path = tmp_path/"mixed.flac"
    path.write_bytes(encode(pcm, 48000, 16, blocksize=4096, plan=plan))
    samples, samplerate = read_flac(path)
    assert samplerate == 48000 and np.array_equal(samples, as_float(pcm, 16))


def test_same_samples_as_the_wav_of_the_clip(tmp_path):
    """FLAC in, WAV in: ShaderAudio sees the same float32 stream either way (ffmpeg's pcm_f32le of both is integer/32768)"""
    pcm = music(3000, 2, 16, seed=9)
    (tmp_path/"a.flac").write_bytes(encode(pcm, 44100, 16, blocksize=576))
    data = pcm.astype("<i2").tobytes()
    (tmp_path/"a.wav").write_bytes(struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(data), b"WAVE", b"fmt ", 16, 1, 2, 44100, 44100*4, 4, 16, b"data", len(data)) + data)
    flac, _ = decode_audio(tmp_path/"a.flac")
    wav, _ = read_wav(tmp_path/"a.wav")
    assert np.array_equal(flac, wav)


def test_many_small_frames_mono_and_unknown_length(tmp_path):
    pcm = music(200*30 + 11, 1, 16, seed=5)                            # 201 frames: two-byte coded frame numbers, a short last block
    path = tmp_path/"mono.flac"
    path.write_bytes(encode(pcm, 22050, 16, blocksize=30, known_length=False))
    assert flac_info(path) == (len(pcm), 1, 22050, 16)                 # STREAMINFO says "unknown": counted by decoding
    samples, _ = read_flac(path)
    assert np.array_equal(samples, as_float(pcm, 16))


def test_corruption_is_detected(tmp_path):
    from shaderflow_amd._native import NativeError
    pcm = music(3000, 2, 16, seed=1)
    stream = bytearray(encode(pcm, 44100, 16, blocksize=1152))
    for position, message in ((len(stream)//2, "CRC"), (50, "CRC|sync|header"), (0, "fLaC")):
        broken = bytearray(stream)
        broken[position] ^= 0x55
        (tmp_path/"broken.flac").write_bytes(bytes(broken))
        with pytest.raises((NativeError, ValueError)):
            read_flac(tmp_path/"broken.flac") if position else decode_audio(tmp_path/"broken.flac")
    (tmp_path/"short.flac").write_bytes(bytes(stream[:len(stream) - 400]))
    with pytest.raises(NativeError):
        read_flac(tmp_path/"short.flac")


def test_probes_answer_flac_files_without_ffprobe(tmp_path):
    from shaderflow_amd.ffmpeg import FFmpeg
    pcm = music(4410, 2, 16)
    path = tmp_path/"probe.flac"
    path.write_bytes(encode(pcm, 44100, 16))
    assert FFmpeg.get_audio_samplerate(path) == 44100 and FFmpeg.get_audio_channels(path) == 2
    assert abs(FFmpeg.get_audio_duration(path) - 0.1) < 1e-9


def test_crc_polynomials_are_the_formats():
    """Independent anchors for the two checksums (catalogue check values): CRC-8 poly 0x07 and CRC-16 poly 0x8005, both MSB first, init 0"""
    from tests.flac_encoder import crc8, crc16
    assert crc8(b"123456789") == 0xF4 and crc16(b"123456789") == 0xFEE8
