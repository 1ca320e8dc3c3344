#!/usr/bin/env python3
"""glsl.hpp unorm8_to_float: fma(c, hi, c*lo) with hi = RN(1/255), lo = RN(1/255 - hi) equals the correctly rounded c/255 for every
byte c — checked in exact rational arithmetic (the float32 product c*lo as the hardware rounds it, then ONE rounding of c*hi + that)."""
from fractions import Fraction

import numpy as np

hi = np.float32(float.fromhex("0x1.010102p-8"))
lo = np.float32(float.fromhex("-0x1.fdfdfep-33"))
assert hi == np.float32(1.0/255.0) and lo == np.float32(1.0/255.0 - float(hi))


def round_to_float32(x: Fraction) -> np.float32:
    f = np.float32(float(x))
    candidates = [f, np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf))]
    return np.float32(min(candidates, key=lambda v: (abs(Fraction(float(v)) - x), int(np.float32(v).view(np.int32)) & 1)))


for c in range(256):
    product = np.float32(np.float32(c)*lo)
    got = round_to_float32(Fraction(c)*Fraction(float(hi)) + Fraction(float(product)))
    want = round_to_float32(Fraction(c, 255))
    assert got == want == np.float32(np.float64(c)/255.0), (c, got, want)
print("fma(c, RN(1/255), c*RN(1/255 - RN(1/255))) == RN(c/255) for all 256 bytes")
