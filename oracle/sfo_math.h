/*
 * ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * sfo_math.h — the "sfmath" specification of the GLSL 3.30 §8 built-ins used by the
 * reference's fragment shaders, restated as plain C on IEEE-754 binary32.
 *
 * GLSL leaves the precision of sin/cos/atan/pow/exp2/log2 to the driver, and the reference
 * (shaderflow/resources/shaders/include/shaderflow.glsl, fragment/ shaders, examples/basic/shaders/ fragments)
 * pins none of it. This file therefore FIXES one
 * admissible implementation: every function below is a finite sequence of correctly-rounded
 * binary32 operations (+ - * / sqrtf fmaf floorf rintf and integer bit moves), evaluated in the
 * written order with no contraction (-ffp-contract=off). The HIP device code restates the same
 * sequences independently (shaderflow_amd/csrc/sfmath.hpp); tests compare both bit for bit, and
 * compare this file against libm in double precision (tests/test_oracle_math.py: ≤ 4 ulp).
 *
 * Polynomial coefficients: Cephes single-precision sinf/cosf/atanf/logf/expf (S. Moshier, public
 * domain) — published constants, restated.
 */
#ifndef SFO_MATH_H
#define SFO_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#define SFO_PI    3.1415926535897932f   /* shaderflow.glsl:7  */
#define SFO_TAU   6.2831853071795864f   /* shaderflow.glsl:8  */

static inline uint32_t sfo_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float sfo_from_bits(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

/* GLSL min/max/clamp/abs/sign: NaN behaviour fixed by the comparison direction written here */
static inline float sfo_min(float a, float b) { return (b < a) ? b : a; }
static inline float sfo_max(float a, float b) { return (a < b) ? b : a; }
static inline float sfo_clamp(float x, float lo, float hi) { return sfo_min(sfo_max(x, lo), hi); }
static inline float sfo_abs(float x) { return sfo_from_bits(sfo_bits(x) & 0x7fffffffu); }
static inline float sfo_floor(float x) { return floorf(x); }
static inline float sfo_fract(float x) { return x - floorf(x); }
static inline float sfo_mod(float x, float y) { return x - y*floorf(x/y); }          /* GLSL 8.3 */
static inline float sfo_mix(float a, float b, float t) { return a*(1.0f - t) + b*t; }   /* GLSL 8.3 */
/* GLSL int(float) on arbitrary values: C leaves NaN and out-of-range conversions undefined (x86 yields INT_MIN, GPUs 0 or
 * saturation); the rule of both implementations is the GPU's: NaN -> 0, saturate */
static inline int32_t sfo_to_int(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}
static inline float sfo_smoothstep(float e0, float e1, float x) {
    float t = sfo_clamp((x - e0)/(e1 - e0), 0.0f, 1.0f);
    return t*t*(3.0f - 2.0f*t);
}
static inline float sfo_sqrt(float x) { return sqrtf(x); }

/* ---- sin / cos: 3-term Cody-Waite reduction by pi/2, Cephes minimax kernels on [-pi/4, pi/4] ---- */

static inline float sfo_sin_kernel(float r) {
    float z = r*r;
    float p = -1.9515295891e-4f;
    p = fmaf(p, z, 8.3321608736e-3f);
    p = fmaf(p, z, -1.6666654611e-1f);
    return fmaf(p*z, r, r);
}
static inline float sfo_cos_kernel(float r) {
    float z = r*r;
    float p = 2.443315711809948e-5f;
    p = fmaf(p, z, -1.388731625493765e-3f);
    p = fmaf(p, z, 4.166664568298827e-2f);
    return fmaf(p*z, z, fmaf(-0.5f, z, 1.0f));
}
static inline float sfo_sincos_reduce(float x, int32_t* quadrant) {
    float k = rintf(x*0x1.45f306p-1f);                 /* x * 2/pi, ties-to-even */
    float r = fmaf(-k, 0x1.921fb6p+0f, x);             /* pi/2 = HI + MID + LO */
    r = fmaf(-k, -0x1.777a5cp-25f, r);
    r = fmaf(-k, -0x1.ee59dap-50f, r);
    /* |k| beyond int32 is outside the supported domain (|x| < 1e9); clamp keeps C defined */
    float kc = sfo_clamp(k, -2147483520.0f, 2147483520.0f);
    *quadrant = (int32_t)kc;
    return r;
}
static inline float sfo_sin(float x) {
    int32_t q; float r = sfo_sincos_reduce(x, &q);
    float v = (q & 1) ? sfo_cos_kernel(r) : sfo_sin_kernel(r);
    return (q & 2) ? -v : v;
}
static inline float sfo_cos(float x) {
    int32_t q; float r = sfo_sincos_reduce(x, &q);
    q += 1;
    float v = (q & 1) ? sfo_cos_kernel(r) : sfo_sin_kernel(r);
    return (q & 2) ? -v : v;
}

/* ---- atan: Cephes atanf kernel with the tan(pi/8) split ---- */

static inline float sfo_atan_unit(float t) {           /* t in [0, 1] */
    float base = 0.0f, u = t;
    if (t > 0x1.a8279ap-2f) {                          /* tan(pi/8) */
        u = (t - 1.0f)/(t + 1.0f);
        base = 0x1.921fb6p-1f;                         /* pi/4 */
    }
    float z = u*u;
    float p = 8.05374449538e-2f;
    p = fmaf(p, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    return base + fmaf(p*z, u, u);
}
/* GLSL atan(y, x) */
static inline float sfo_atan2(float y, float x) {
    float ax = sfo_abs(x), ay = sfo_abs(y);
    float mx = sfo_max(ax, ay), mn = sfo_min(ax, ay);
    float t = (mx == 0.0f) ? 0.0f : mn/mx;
    float a = sfo_atan_unit(t);
    if (ay > ax) a = 0x1.921fb6p+0f - a;               /* pi/2 - a */
    if (x < 0.0f) a = 0x1.921fb6p+1f - a;              /* pi - a   */
    return (y < 0.0f) ? -a : a;
}
/* GLSL atan(y_over_x) */
static inline float sfo_atan(float v) {
    float av = sfo_abs(v);
    float a = (av > 1.0f) ? (0x1.921fb6p+0f - sfo_atan_unit(1.0f/av)) : sfo_atan_unit(av);
    return (v < 0.0f) ? -a : a;
}

/* ---- log2 / exp2 / pow / exp ---- */

static inline float sfo_log2(float x) {
    if (x < 0.0f || x != x) return sfo_from_bits(0x7fc00000u);      /* NaN */
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return INFINITY;
    uint32_t u = sfo_bits(x);
    int32_t e = 0;
    if (u < 0x00800000u) { x = x*0x1p+23f; u = sfo_bits(x); e = -23; }  /* subnormal */
    e += (int32_t)(u >> 23) - 127;
    float m = sfo_from_bits((u & 0x007fffffu) | 0x3f800000u);      /* [1, 2) */
    if (m > 0x1.6a09e6p+0f) { m = m*0.5f; e += 1; }                 /* → [sqrt(.5), sqrt(2)) */
    float f = m - 1.0f;
    float z = f*f;
    float p = 7.0376836292e-2f;
    p = fmaf(p, f, -1.1514610310e-1f);
    p = fmaf(p, f, 1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f, 1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f, 2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f, 3.3333331174e-1f);
    float ln = fmaf(p*z, f, fmaf(-0.5f, z, f));                    /* ln(m) */
    return fmaf(ln, 0x1.715476p+0f, (float)e);                     /* ln(m)*log2(e) + e */
}
static inline float sfo_exp2(float x) {
    if (x != x) return x;
    if (x >= 128.0f) return INFINITY;
    if (x < -150.0f) return 0.0f;
    float n = rintf(x);
    float f = (x - n)*0x1.62e430p-1f;                              /* (x-n)*ln2, |f| <= 0.3466 */
    float p = 1.9875691500e-4f;
    p = fmaf(p, f, 1.3981999507e-3f);
    p = fmaf(p, f, 8.3334519073e-3f);
    p = fmaf(p, f, 4.1665795894e-2f);
    p = fmaf(p, f, 1.6666665459e-1f);
    p = fmaf(p, f, 5.0000001201e-1f);
    float r = fmaf(p*f, f, f) + 1.0f;                              /* e^f */
    int32_t ni = (int32_t)n;
    /* scale by 2^ni in two exact steps so that subnormal results round once */
    int32_t n1 = ni/2, n2 = ni - n1;
    r = r*sfo_from_bits((uint32_t)(n1 + 127) << 23);
    return r*sfo_from_bits((uint32_t)(n2 + 127) << 23);
}
/* GLSL pow(x, y) = exp2(y*log2(x)); undefined for x < 0 → NaN here; pow(0, y>0) = 0 */
static inline float sfo_pow(float x, float y) {
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : INFINITY);
    return sfo_exp2(y*sfo_log2(x));
}
static inline float sfo_exp(float x) { return sfo_exp2(x*0x1.715476p+0f); }
static inline float sfo_log(float x) { return sfo_log2(x)*0x1.62e430p-1f; }   /* ln 2 */

#endif
