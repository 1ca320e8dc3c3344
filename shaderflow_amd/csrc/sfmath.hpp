// sfmath.hpp — GLSL 3.30 §8 built-in functions for the gfx950 fragment kernels.
//
// GLSL leaves sin/cos/atan/pow/exp2/log2 precision to the driver; the reference (include/shaderflow.glsl
// and the example fragments) runs on whatever OpenGL driver is present. Here every built-in is a fixed
// sequence of correctly rounded binary32 operations (add, mul, div, sqrt, fma, floor, rint, bit moves) so
// that a frame is a pure function of its inputs on any CDNA4 part, and so that the CPU oracle can check
// it bit for bit. No v_sin/v_cos/v_exp/v_log (≈1 ulp, unspecified), no contraction (-ffp-contract=off):
// where a fused multiply-add is wanted it is written as fmaf().
//
// Kernels: Cephes single-precision minimax polynomials (S. Moshier, public domain) with a three-term
// Cody-Waite reduction by pi/2 done in fma (the product k*HI never rounds, so the reduced argument is
// exact to one rounding for |x| well beyond any iTime a scene reaches).
#pragma once

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

#define SF_HD __host__ __device__ __forceinline__

namespace sf {

constexpr float PI = 3.1415926535897932f;      // shaderflow.glsl:7
constexpr float TAU = 6.2831853071795864f;     // shaderflow.glsl:8
constexpr float HALF_PI = 0x1.921fb6p+0f;
constexpr float QUARTER_PI = 0x1.921fb6p-1f;

SF_HD uint32_t f2u(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(x);
#else
    uint32_t u; __builtin_memcpy(&u, &x, 4); return u;
#endif
}
SF_HD float u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float x; __builtin_memcpy(&x, &u, 4); return x;
#endif
}

// Comparison direction decides what a NaN operand does; both sides of the parity check use these.
SF_HD float min(float a, float b) { return (b < a) ? b : a; }
SF_HD float max(float a, float b) { return (a < b) ? b : a; }
SF_HD float clamp(float x, float lo, float hi) { return sf::min(sf::max(x, lo), hi); }
SF_HD float abs(float x) { return u2f(f2u(x) & 0x7fffffffu); }
SF_HD float floor(float x) { return ::floorf(x); }
SF_HD float fract(float x) { return x - ::floorf(x); }
SF_HD float mod(float x, float y) { return x - y*::floorf(x/y); }
SF_HD float mix(float a, float b, float t) { return a*(1.0f - t) + b*t; }
SF_HD float sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
SF_HD float sqrt(float x) { return ::sqrtf(x); }
// GLSL int(float) where the value can be anything (NaN out of a diverging iteration, huge products): C leaves those
// conversions undefined and CPUs and GPUs disagree, so both implementations fix the GPU's rule: NaN -> 0, saturate
SF_HD int to_int(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return -2147483647 - 1;
    return (int)x;
}
SF_HD float smoothstep(float e0, float e1, float x) {
    float t = sf::clamp((x - e0)/(e1 - e0), 0.0f, 1.0f);
    return t*t*(3.0f - 2.0f*t);
}

// ---- sin / cos --------------------------------------------------------------------------------
SF_HD float sin_poly(float r) {
    float z = r*r;
    float p = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = fmaf(p, z, -1.6666654611e-1f);
    return fmaf(p*z, r, r);
}
SF_HD float cos_poly(float r) {
    float z = r*r;
    float p = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = fmaf(p, z, 4.166664568298827e-2f);
    return fmaf(p*z, z, fmaf(-0.5f, z, 1.0f));
}
struct Reduced { float r; int q; };
SF_HD Reduced reduce_half_pi(float x) {
    float k = ::rintf(x*0x1.45f306p-1f);
    float r = fmaf(-k, 0x1.921fb6p+0f, x);
    r = fmaf(-k, -0x1.777a5cp-25f, r);
    r = fmaf(-k, -0x1.ee59dap-50f, r);
    float kc = sf::clamp(k, -2147483520.0f, 2147483520.0f);
    return {r, (int)kc};
}
SF_HD float quadrant_value(float r, int q) {
    float v = (q & 1) ? cos_poly(r) : sin_poly(r);
    return (q & 2) ? -v : v;
}
SF_HD float sin(float x) { Reduced a = reduce_half_pi(x); return quadrant_value(a.r, a.q); }
SF_HD float cos(float x) { Reduced a = reduce_half_pi(x); return quadrant_value(a.r, a.q + 1); }

// ---- atan -------------------------------------------------------------------------------------
SF_HD float atan01(float t) {
    bool upper = t > 0x1.a8279ap-2f;               // tan(pi/8)
    float u = upper ? (t - 1.0f)/(t + 1.0f) : t;
    float base = upper ? QUARTER_PI : 0.0f;
    float z = u*u;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    return base + fmaf(p*z, u, u);
}
SF_HD float atan(float y, float x) {               // GLSL atan(y, x)
    float ax = sf::abs(x), ay = sf::abs(y);
    float hi = sf::max(ax, ay), lo = sf::min(ax, ay);
    float t = (hi == 0.0f) ? 0.0f : lo/hi;
    float a = atan01(t);
    if (ay > ax) a = HALF_PI - a;
    if (x < 0.0f) a = 0x1.921fb6p+1f - a;
    return (y < 0.0f) ? -a : a;
}
SF_HD float atan(float v) {                        // GLSL atan(y_over_x)
    float av = sf::abs(v);
    float a = (av > 1.0f) ? (HALF_PI - atan01(1.0f/av)) : atan01(av);
    return (v < 0.0f) ? -a : a;
}

// ---- log2 / exp2 / pow / exp ------------------------------------------------------------------
SF_HD float log2(float x) {
    if (x < 0.0f || x != x) return u2f(0x7fc00000u);
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return INFINITY;
    uint32_t u = f2u(x);
    int e = 0;
    if (u < 0x00800000u) { x = x*0x1p+23f; u = f2u(x); e = -23; }
    e += (int)(u >> 23) - 127;
    float m = u2f((u & 0x007fffffu) | 0x3f800000u);
    if (m > 0x1.6a09e6p+0f) { m = m*0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f*f;
    float p = fmaf(7.0376836292e-2f, f, -1.1514610310e-1f);
    p = fmaf(p, f, 1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f, 1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f, 2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f, 3.3333331174e-1f);
    float ln = fmaf(p*z, f, fmaf(-0.5f, z, f));
    return fmaf(ln, 0x1.715476p+0f, (float)e);
}
SF_HD float exp2(float x) {
    if (x != x) return x;
    if (x >= 128.0f) return INFINITY;
    if (x < -150.0f) return 0.0f;
    float n = ::rintf(x);
    float f = (x - n)*0x1.62e430p-1f;
    float p = fmaf(1.9875691500e-4f, f, 1.3981999507e-3f);
    p = fmaf(p, f, 8.3334519073e-3f);
    p = fmaf(p, f, 4.1665795894e-2f);
    p = fmaf(p, f, 1.6666665459e-1f);
    p = fmaf(p, f, 5.0000001201e-1f);
    float r = fmaf(p*f, f, f) + 1.0f;
    int ni = (int)n;
    int n1 = ni/2, n2 = ni - n1;
    r = r*u2f((uint32_t)(n1 + 127) << 23);
    return r*u2f((uint32_t)(n2 + 127) << 23);
}
SF_HD float pow(float x, float y) {
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : INFINITY);
    return sf::exp2(y*sf::log2(x));
}
SF_HD float exp(float x) { return sf::exp2(x*0x1.715476p+0f); }
SF_HD float log(float x) { return sf::log2(x)*0x1.62e430p-1f; }

}  // namespace sf
