#!/usr/bin/env python3
"""How far the speculated polar angle of visualizer_fast_post (v_rcp_f32 instead of the two IEEE divisions, a multiplication by
1/PI instead of the third) can be from the generic chain's (sfmath.hpp atan / glsl.hpp atan1n): float32 emulation over 4 M points,
the hardware reciprocals modelled as the correctly rounded ones moved by -1, 0 or +1 ulp at random. The kernel's margin
(height*1e-6 + 2e-6 on circle*height) assumes |difference| <= 1e-6; this prints the largest one seen (1.8e-7)."""
import numpy as np

rng = np.random.default_rng(1)
f = np.float32
N = 4_000_000
x = rng.uniform(-3, 3, N).astype(f)
y = rng.uniform(-2, 2, N).astype(f)
x[:N//4] *= f(0.05)
y[:N//4] *= f(0.05)


def within_an_ulp(v):
    return (v.view(np.int32) + rng.integers(-1, 2, v.shape).astype(np.int32)).view(f)


T8 = f(float.fromhex("0x1.a8279ap-2"))
QPI, HPI, PI = f(np.pi/4), f(np.pi/2), f(float.fromhex("0x1.921fb6p+1"))


def fma(a, b, c):
    return (a.astype(np.float64)*b.astype(np.float64) + c.astype(np.float64)).astype(f)


def poly(u):
    z = u*u
    p = fma(np.full_like(z, f(8.05374449538e-2)), z, np.full_like(z, f(-1.38776856032e-1)))
    p = fma(p, z, np.full_like(z, f(1.99777106478e-1)))
    p = fma(p, z, np.full_like(z, f(-3.33329491539e-1)))
    return fma(p*z, u, u)


ax, ay = np.abs(x), np.abs(y)
hi, lo = np.maximum(ax, ay), np.minimum(ax, ay)
# the generic chain
t = (lo.astype(np.float64)/hi.astype(np.float64)).astype(f)
upper = t > T8
u = np.where(upper, ((t - f(1)).astype(np.float64)/(t + f(1)).astype(np.float64)).astype(f), t)
a = np.where(upper, QPI, f(0)) + poly(u)
a = np.where(ay > ax, HPI - a, a)
a = np.where(x < 0, PI - a, a)
circle = (a.astype(np.float64)/np.float64(PI)).astype(f)
# the speculated one
t2 = lo*within_an_ulp((f(1)/hi).astype(f))
upper2 = t2 > T8
u2 = np.where(upper2, (t2 - f(1))*within_an_ulp((f(1)/(t2 + f(1))).astype(f)), t2)
a2 = np.where(upper2, QPI, f(0)) + poly(u2)
a2 = np.where(ay > ax, HPI - a2, a2)
a2 = np.where(x < 0, PI - a2, a2)
circle2 = a2*f(float.fromhex("0x1.45f306p-2"))
d = np.abs(circle2.astype(np.float64) - circle.astype(np.float64))
print(f"max |speculated - exact| of atan(y, x)/PI over {N} points: {d.max():.3e}   (99.99 %: {np.quantile(d, 0.9999):.3e})")
assert d.max() <= 1.0e-6
