"""
A small FLAC ENCODER, test infrastructure only: the product decodes FLAC natively (csrc/flac.inc) and this image has no encoder
(no ffmpeg, no flac binary, no soundfile), so the test streams are written here, straight from the format description
(RFC 9639 / xiph.org/flac/format.html): STREAMINFO, frames with CRC-8/CRC-16, CONSTANT / VERBATIM / FIXED / LPC subframes, Rice
partitions (both parameter widths, escape partitions), wasted bits, the four channel assignments. Slow, readable, exhaustive in
the features it can switch on — not a compressor.
"""
from __future__ import annotations

import numpy as np


class BitWriter:
    def __init__(self):
        self.bits: list[int] = []

    def write(self, value: int, n: int) -> None:
        value &= (1 << n) - 1
        self.bits.extend((value >> (n - 1 - k)) & 1 for k in range(n))

    def unary(self, zeros: int) -> None:
        self.bits.extend([0]*zeros + [1])

    def align(self) -> None:
        self.bits.extend([0]*((-len(self.bits)) % 8))

    def bytes(self) -> bytes:
        assert len(self.bits) % 8 == 0
        return np.packbits(np.array(self.bits, np.uint8)).tobytes()


def crc8(data: bytes) -> int:
    crc = 0
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = ((crc << 1) ^ 0x07) & 0xff if crc & 0x80 else (crc << 1) & 0xff
    return crc


def crc16(data: bytes) -> int:
    crc = 0
    for byte in data:
        crc ^= byte << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x8005) & 0xffff if crc & 0x8000 else (crc << 1) & 0xffff
    return crc


def utf8_number(value: int) -> bytes:
    if value < 0x80:
        return bytes([value])
    out, lead_bits = [], 6
    while value >= (1 << lead_bits):
        out.append(0x80 | (value & 0x3f)); value >>= 6; lead_bits -= 1
    n = len(out) + 1
    return bytes([((0xff << (8 - n)) & 0xff) | value] + out[::-1])


def rice(w: BitWriter, residual: list[int], order: int, blocksize: int, partition_order: int, wide: bool, escape_partitions=()) -> None:
    """RESIDUAL: coding method, partition order, then per partition a Rice parameter chosen for its mean magnitude"""
    w.write(1 if wide else 0, 2)
    w.write(partition_order, 4)
    index = 0
    for p in range(1 << partition_order):
        count = (blocksize >> partition_order) - (order if p == 0 else 0)
        part = residual[index:index + count]; index += count
        if p in escape_partitions:
            raw = max(1, max((abs(v) for v in part), default=0).bit_length() + 1)
            w.write(31 if wide else 15, 5 if wide else 4)
            w.write(raw, 5)
            for v in part:
                w.write(v, raw)
            continue
        mean = (sum(abs(v) for v in part)/max(1, len(part)))
        k = min(14, max(0, int(mean).bit_length()))
        w.write(k, 5 if wide else 4)
        for v in part:
            u = (v << 1) if v >= 0 else ((-v) << 1) - 1
            w.unary(u >> k)
            w.write(u & ((1 << k) - 1), k)
    assert index == len(residual)


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def subframe(w: BitWriter, samples: list[int], bps: int, kind: str, order: int = 0, lpc=None, partition_order: int = 0, wide: bool = False,
             escape_partitions=(), wasted: int = 0) -> None:
    """kind: constant | verbatim | fixed | lpc.  lpc = (coefficients, precision, shift)"""
    n = len(samples)
    w.write(0, 1)
    code = {"constant": 0, "verbatim": 1, "fixed": 8 + order, "lpc": 32 + (order - 1)}[kind]
    w.write(code, 6)
    if wasted:
        assert all(v % (1 << wasted) == 0 for v in samples)
        w.write(1, 1); w.unary(wasted - 1)
        samples = [v >> wasted for v in samples]; bps -= wasted
    else:
        w.write(0, 1)
    if kind == "constant":
        assert len(set(samples)) == 1
        w.write(samples[0], bps)
    elif kind == "verbatim":
        for v in samples:
            w.write(v, bps)
    else:
        coefficients, shift = (FIXED[order], 0) if kind == "fixed" else (lpc[0], lpc[2])
        for v in samples[:order]:
            w.write(v, bps)
        if kind == "lpc":
            w.write(lpc[1] - 1, 4); w.write(shift, 5)
            for c in coefficients:
                w.write(c, lpc[1])
        residual = [samples[i] - (sum(c*samples[i - 1 - j] for j, c in enumerate(coefficients)) >> shift) for i in range(order, n)]
        rice(w, residual, order, n, partition_order, wide, escape_partitions)


def encode(pcm: np.ndarray, samplerate: int, bits: int, blocksize: int = 1152, plan=None, known_length: bool = True) -> bytes:
    """pcm: (n, channels) integer samples of `bits` bits. plan(frame_index, channels) → dict(assignment=0..10 or None for independent,
    subframes=[kwargs for subframe()] per coded channel); default: FIXED order 2, independent channels"""
    pcm = np.asarray(pcm, np.int64)
    n, channels = pcm.shape
    info = BitWriter()
    info.write(blocksize, 16); info.write(blocksize, 16); info.write(0, 24); info.write(0, 24)
    info.write(samplerate, 20); info.write(channels - 1, 3); info.write(bits - 1, 5); info.write(n if known_length else 0, 36)
    info.write(0, 128)                                                # MD5 "not computed"
    out = bytearray(b"fLaC" + bytes([0x80 | 0]) + (34).to_bytes(3, "big") + info.bytes())
    size_codes = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
    rate_codes = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
    bits_codes = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}
    for frame, first in enumerate(range(0, n, blocksize)):
        block = pcm[first:first + blocksize]
        count = len(block)
        choice = plan(frame, channels) if plan else {}
        assignment = choice.get("assignment")
        w = BitWriter()
        w.write(0x3ffe, 14); w.write(0, 1); w.write(0, 1)             # sync, reserved, fixed-blocksize stream (frame number coded)
        size_code = size_codes.get(count)
        if size_code is None:
            size_code = 6 if count <= 256 else 7
        w.write(size_code, 4)
        rate_code = rate_codes.get(samplerate, 13 if samplerate < 65536 else 0)
        w.write(rate_code, 4)
        w.write(assignment if assignment is not None else channels - 1, 4)
        w.write(bits_codes[bits], 3); w.write(0, 1)
        for byte in utf8_number(frame):
            w.write(byte, 8)
        if size_code == 6:
            w.write(count - 1, 8)
        elif size_code == 7:
            w.write(count - 1, 16)
        if rate_code == 13:
            w.write(samplerate, 16)
        w.write(crc8(w.bytes()), 8)
        left, right = (block[:, 0], block[:, 1]) if channels == 2 else (None, None)
        if assignment == 8:
            coded = [(left, bits), (left - right, bits + 1)]
        elif assignment == 9:
            coded = [(left - right, bits + 1), (right, bits)]
        elif assignment == 10:
            coded = [((left + right) >> 1, bits), (left - right, bits + 1)]
        else:
            coded = [(block[:, c], bits) for c in range(channels)]
        subframes = choice.get("subframes") or [dict(kind="fixed", order=min(2, count))]*len(coded)
        for (values, width), kw in zip(coded, subframes):
            subframe(w, [int(v) for v in values], width, **kw)
        w.align()
        w.write(crc16(w.bytes()), 16)
        out += w.bytes()
    return bytes(out)
