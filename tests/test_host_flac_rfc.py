"""
Independent evidence for the native FLAC decoder (csrc/flac.inc): the three example streams RFC 9639 prints byte by
byte in its Appendix D ("Examples": a one-sample stereo stream with wasted bits; a 19-sample stereo stream in two frames with a
SEEKTABLE, a VORBIS_COMMENT written by "reference libFLAC 1.3.3" and a PADDING block, left-side stereo, FIXED and LPC-free
predictors, partitioned Rice coding with an escape; a 24-sample mono 8-bit stream with an order-2 FIXED predictor…). They are spec
text — produced by the reference FLAC encoder, not by tests/flac_encoder.py — and they carry their own answer: STREAMINFO holds the
MD5 of the unencoded audio as the ENCODER computed it, every frame header a CRC-8 and every frame a CRC-16 (both checked by the
decoder). The decoder must reproduce the sample values the RFC walks through and the MD5 must match. CPU only.
"""
import hashlib

import numpy as np
import pytest

from shaderflow_amd.audio.reader import flac_info, read_flac

EXAMPLES = {
    # RFC 9639 D.1: "Decoding Example 1" — 1 sample, 2 channels, 16 bit, 44.1 kHz; verbatim subframes with 2 and 4 wasted bits
    "D.1": ("""664c 6143 8000 0022 1000 1000 0000 0f00 000f 0ac4 42f0 0000 0001 3e84 b418 07dc 6903 0758
               6a3d ad1a 2e0f fff8 6918 0000 bf03 58fd 0312 8baa 9a""",
            (1, 2, 44100, 16), [[25588], [10416]]),
    # D.2: 19 samples in frames of 16 and 3; STREAMINFO + SEEKTABLE + VORBIS_COMMENT + PADDING
    "D.2": ("""664c 6143 0000 0022 0010 0010 0000 1700 0044 0ac4 42f0 0000 0013 d5b0 5649 75e9 8b8d 8b93
               0422 757b 8103 0300 0012 0000 0000 0000 0000 0000 0000 0000 0000 0010 0400 003a 2000 0000
               7265 6665 7265 6e63 6520 6c69 6246 4c41 4320 312e 332e 3320 3230 3139 3038 3034 0100 0000
               0e00 0000 5449 544c 453d d7a9 d79c d795 d79d 8100 0006 0000 0000 0000 fff8 6998 000f 9912
               0867 0162 3d14 4299 8f5d f70d 6fe0 0c17 caeb 2100 0ee7 a77a 24a1 590c 1217 b603 097b 784f
               aa9a 33d2 85e0 70ad 5b1b 4851 b401 0d99 d2cd 1a68 f1e6 b810 fff8 6918 0102 a402 c382 c40b
               c14a 03ee 48dd 03b6 7c13 30""",
            (19, 2, 44100, 16),
            [[10372, 18041, 14942, 17876, 15627, 17899, 16242, 18077, 16824, 18263, 17295, -14418, -15201, -14508, -15195, -14818, -15486, -15349, -16054],
             [6070, 10545, 8743, 10449, 9143, 10463, 9502, 10569, 9840, 10680, 10113, -8428, -8895, -8476, -8896, -8653, -9072, -8958, -9410]]),
    # D.3: 24 samples, mono, 8 bit, 32 kHz
    "D.3": ("""664c 6143 8000 0022 1000 1000 0000 1f00 001f 07d0 0070 0000 0018 f8f9 e396 f5cb cfc6 dc80
               7f99 7790 6b32 fff8 6802 0017 e944 004f 6f31 3d10 47d2 27cb 6d09 0831 452b dc28 2222 8057 a3""",
            (24, 1, 32000, 8),
            [[0, 79, 111, 78, 8, -61, -90, -68, -13, 42, 67, 53, 13, -27, -46, -38, -12, 14, 24, 19, 6, -4, -5, 0]]),
}


@pytest.mark.parametrize("name", list(EXAMPLES))
def test_rfc_9639_appendix_d_streams(tmp_path, name):
    text, info, channels = EXAMPLES[name]
    data = bytes.fromhex("".join(text.split()))
    path = tmp_path/f"{name}.flac"
    path.write_bytes(data)
    assert flac_info(path) == info
    samples, samplerate = read_flac(path)
    bits = info[3]
    integers = np.rint(samples.astype(np.float64)*(1 << (bits - 1))).astype(np.int64)
    assert samplerate == info[2] and integers.T.tolist() == channels
    # the MD5 of the interleaved little-endian PCM, as computed by the encoder that wrote the stream (STREAMINFO bytes 18-33)
    pcm = integers.astype({8: "<i1", 16: "<i2"}[bits]).tobytes()
    assert hashlib.md5(pcm).digest() == data[26:42]


def test_a_flipped_bit_in_an_rfc_stream_is_caught_by_its_crc(tmp_path):
    data = bytearray(bytes.fromhex("".join(EXAMPLES["D.3"][0].split())))
    data[-10] ^= 0x04
    path = tmp_path/"broken.flac"
    path.write_bytes(bytes(data))
    with pytest.raises(Exception, match="(?i)crc|flac"):
        read_flac(path)
