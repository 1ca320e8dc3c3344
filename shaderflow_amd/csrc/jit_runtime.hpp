// jit_runtime.hpp — GLSL 3.30 as C++: what a run-time translated fragment (shaderflow_amd/glsl2hip.py) is compiled against.
//
// The reference hands the assembled GLSL to the OpenGL driver (shader.py:190-239, 313-349: `opengl.program(vertex_shader,
// fragment_shader)`). Here a fragment that is not in the registry of restated kernels is rewritten token by token into the
// body of a C++ struct (uniforms, varyings and globals become members, functions become member functions), compiled with
// hipcc for gfx950 against this header and loaded as a code object (capi: sfx_program_load). This header supplies
//   * the vector / matrix types with swizzles (GLSL 3.30 §4.1, §5.5, §5.9-5.10),
//   * the built-in functions (§8) on the deterministic binary32 routines of sfmath.hpp — the same operations the
//     restated kernels use, so a translated fragment and its hand-written twin produce the same bits,
//   * the sampler functions on glsl.hpp's texture unit (§8.7, OpenGL 3.3 §3.8),
//   * the reference's prelude API: include/shaderflow.glsl:1-472, include/camera.glsl:1-157, include/complex.glsl:1-63
//     (constants, coordinate conversions, textures, palettes, SDFs, colour, noise, `GetCamera`),
//   * FragmentBase (built-in uniforms of scene.py:687-703 / camera.py:196-201 / audio modules, the varyings of
//     vertex/default.glsl:1-17) and the kernel entry points a code object exports.
#pragma once

#include "render_kernels.hpp"
#include "jit_swizzles.inc"

#include <type_traits>

namespace sf { namespace rt {

typedef unsigned int uint;
struct vec2; struct vec3; struct vec4;

// ---- swizzles ---------------------------------------------------------------------------------------------------------
// A swizzle is a member of the vector's anonymous union (same storage): reading converts it to a vector of the selected
// components, writing assigns them (GLSL requires distinct components for that; not checked).
template <class T, class V, int P, int... I>
struct SwzT {
    T d[P];
    SF_HD operator V() const { return V(d[I]...); }
    SF_HD SwzT& operator=(const V& v) { int k = 0; ((d[I] = v.d[k++]), ...); return *this; }
    SF_HD SwzT& operator=(const SwzT& o) { const V v = o; return *this = v; }
    // `v.xy op= x` for anything `vec2 op x` is defined for (vectors, scalars, matrices)
    template <class X> SF_HD SwzT& operator+=(const X& x) { const V me = *this; return *this = me + x; }
    template <class X> SF_HD SwzT& operator-=(const X& x) { const V me = *this; return *this = me - x; }
    template <class X> SF_HD SwzT& operator*=(const X& x) { const V me = *this; return *this = me*x; }
    template <class X> SF_HD SwzT& operator/=(const X& x) { const V me = *this; return *this = me/x; }
    template <class X> SF_HD SwzT& operator%=(const X& x) { const V me = *this; return *this = me % x; }
    template <class X> SF_HD SwzT& operator&=(const X& x) { const V me = *this; return *this = me & x; }
    template <class X> SF_HD SwzT& operator|=(const X& x) { const V me = *this; return *this = me | x; }
    template <class X> SF_HD SwzT& operator^=(const X& x) { const V me = *this; return *this = me ^ x; }
    template <class X> SF_HD SwzT& operator<<=(const X& x) { const V me = *this; return *this = me << x; }
    template <class X> SF_HD SwzT& operator>>=(const X& x) { const V me = *this; return *this = me >> x; }
    SF_HD T operator[](int k) const { const int index[] = {I...}; return d[index[k]]; }
};
template <class V, int P, int... I> using Swz = SwzT<float, V, P, I...>;

#define SF_RT_COMMON(V, N) \
    SF_HD V& operator=(const V& o) { for (int k = 0; k < N; k++) d[k] = o.d[k]; return *this; } \
    SF_HD V(const V& o) { for (int k = 0; k < N; k++) d[k] = o.d[k]; } \
    SF_HD float& operator[](int k) { return d[k]; } \
    SF_HD float operator[](int k) const { return d[k]; } \
    SF_HD V& operator+=(const V& o) { for (int k = 0; k < N; k++) d[k] = d[k] + o.d[k]; return *this; } \
    SF_HD V& operator-=(const V& o) { for (int k = 0; k < N; k++) d[k] = d[k] - o.d[k]; return *this; } \
    SF_HD V& operator*=(const V& o) { for (int k = 0; k < N; k++) d[k] = d[k]*o.d[k]; return *this; } \
    SF_HD V& operator/=(const V& o) { for (int k = 0; k < N; k++) d[k] = d[k]/o.d[k]; return *this; } \
    SF_HD V& operator+=(float s) { for (int k = 0; k < N; k++) d[k] = d[k] + s; return *this; } \
    SF_HD V& operator-=(float s) { for (int k = 0; k < N; k++) d[k] = d[k] - s; return *this; } \
    SF_HD V& operator*=(float s) { for (int k = 0; k < N; k++) d[k] = d[k]*s; return *this; } \
    SF_HD V& operator/=(float s) { for (int k = 0; k < N; k++) d[k] = d[k]/s; return *this; }

// integer vectors: same shape as the float ones (names, array and swizzles in one union), for int and for uint
struct ivec2; struct ivec3; struct ivec4; struct uvec2; struct uvec3; struct uvec4;
#define SF_RT_T int
#define SF_RT_V2 ivec2
#define SF_RT_V3 ivec3
#define SF_RT_V4 ivec4
#define SF_RT_O2 uvec2
#define SF_RT_O3 uvec3
#define SF_RT_O4 uvec4
#include "jit_intvec.inc"
#undef SF_RT_T
#undef SF_RT_V2
#undef SF_RT_V3
#undef SF_RT_V4
#undef SF_RT_O2
#undef SF_RT_O3
#undef SF_RT_O4
#define SF_RT_T uint
#define SF_RT_V2 uvec2
#define SF_RT_V3 uvec3
#define SF_RT_V4 uvec4
#define SF_RT_O2 ivec2
#define SF_RT_O3 ivec3
#define SF_RT_O4 ivec4
#include "jit_intvec.inc"
#undef SF_RT_T
#undef SF_RT_V2
#undef SF_RT_V3
#undef SF_RT_V4
#undef SF_RT_O2
#undef SF_RT_O3
#undef SF_RT_O4
struct bvec2 { bool x, y; };
struct bvec3 { bool x, y, z; };
struct bvec4 { bool x, y, z, w; };

struct vec2 {
    union {
        struct { float x, y; };
        struct { float r, g; };
        struct { float s, t; };
        float d[2];
#define SF_RT_A(n, ...) Swz<vec2, 2, __VA_ARGS__> n;
#define SF_RT_B(n, ...) Swz<vec3, 2, __VA_ARGS__> n;
#define SF_RT_C(n, ...) Swz<vec4, 2, __VA_ARGS__> n;
        SF_SWIZZLES_2(SF_RT_A, SF_RT_B, SF_RT_C)
#undef SF_RT_A
#undef SF_RT_B
#undef SF_RT_C
    };
    SF_HD vec2() { x = 0.0f; y = 0.0f; }
    SF_HD explicit vec2(float v) { x = v; y = v; }
    SF_HD vec2(float a, float b) { x = a; y = b; }
    SF_HD vec2(const ivec2& v) { x = (float)v.x; y = (float)v.y; }      // GLSL converts ivec/uvec to vec implicitly (§4.1.10)
    SF_HD vec2(const uvec2& v) { x = (float)v.x; y = (float)v.y; }
    SF_HD explicit vec2(const bvec2& v) { x = v.x ? 1.0f : 0.0f; y = v.y ? 1.0f : 0.0f; }
    SF_HD explicit vec2(const vec3& v);
    SF_HD explicit vec2(const vec4& v);
    SF_RT_COMMON(vec2, 2)
};
struct vec3 {
    union {
        struct { float x, y, z; };
        struct { float r, g, b; };
        struct { float s, t, p; };
        float d[3];
#define SF_RT_A(n, ...) Swz<vec2, 3, __VA_ARGS__> n;
#define SF_RT_B(n, ...) Swz<vec3, 3, __VA_ARGS__> n;
#define SF_RT_C(n, ...) Swz<vec4, 3, __VA_ARGS__> n;
        SF_SWIZZLES_3(SF_RT_A, SF_RT_B, SF_RT_C)
#undef SF_RT_A
#undef SF_RT_B
#undef SF_RT_C
    };
    SF_HD vec3() { x = 0.0f; y = 0.0f; z = 0.0f; }
    SF_HD explicit vec3(float v) { x = v; y = v; z = v; }
    SF_HD vec3(float a, float b, float c) { x = a; y = b; z = c; }
    SF_HD vec3(const vec2& a, float c) { x = a.x; y = a.y; z = c; }
    SF_HD vec3(float a, const vec2& b) { x = a; y = b.x; z = b.y; }
    SF_HD vec3(const ivec3& v) { x = (float)v.x; y = (float)v.y; z = (float)v.z; }
    SF_HD vec3(const uvec3& v) { x = (float)v.x; y = (float)v.y; z = (float)v.z; }
    SF_HD explicit vec3(const bvec3& v) { x = v.x ? 1.0f : 0.0f; y = v.y ? 1.0f : 0.0f; z = v.z ? 1.0f : 0.0f; }
    SF_HD explicit vec3(const vec4& v);
    SF_RT_COMMON(vec3, 3)
};
struct vec4 {
    union {
        struct { float x, y, z, w; };
        struct { float r, g, b, a; };
        struct { float s, t, p, q; };
        float d[4];
#define SF_RT_A(n, ...) Swz<vec2, 4, __VA_ARGS__> n;
#define SF_RT_B(n, ...) Swz<vec3, 4, __VA_ARGS__> n;
#define SF_RT_C(n, ...) Swz<vec4, 4, __VA_ARGS__> n;
        SF_SWIZZLES_4(SF_RT_A, SF_RT_B, SF_RT_C)
#undef SF_RT_A
#undef SF_RT_B
#undef SF_RT_C
    };
    SF_HD vec4() { x = 0.0f; y = 0.0f; z = 0.0f; w = 0.0f; }
    SF_HD explicit vec4(float v) { x = v; y = v; z = v; w = v; }
    SF_HD vec4(float a, float b, float c, float e) { x = a; y = b; z = c; w = e; }
    SF_HD vec4(const vec3& v, float e) { x = v.x; y = v.y; z = v.z; w = e; }
    SF_HD vec4(float a, const vec3& v) { x = a; y = v.x; z = v.y; w = v.z; }
    SF_HD vec4(const vec2& u, const vec2& v) { x = u.x; y = u.y; z = v.x; w = v.y; }
    SF_HD vec4(const vec2& u, float c, float e) { x = u.x; y = u.y; z = c; w = e; }
    SF_HD vec4(float a, const vec2& u, float e) { x = a; y = u.x; z = u.y; w = e; }
    SF_HD vec4(float a, float b, const vec2& u) { x = a; y = b; z = u.x; w = u.y; }
    SF_HD vec4(const ivec4& v) { x = (float)v.x; y = (float)v.y; z = (float)v.z; w = (float)v.w; }
    SF_HD vec4(const uvec4& v) { x = (float)v.x; y = (float)v.y; z = (float)v.z; w = (float)v.w; }
    SF_HD explicit vec4(const bvec4& v) { x = v.x ? 1.0f : 0.0f; y = v.y ? 1.0f : 0.0f; z = v.z ? 1.0f : 0.0f; w = v.w ? 1.0f : 0.0f; }
    SF_RT_COMMON(vec4, 4)
};
SF_HD vec2::vec2(const vec3& v) { x = v.x; y = v.y; }
SF_HD vec2::vec2(const vec4& v) { x = v.x; y = v.y; }
SF_HD vec3::vec3(const vec4& v) { x = v.x; y = v.y; z = v.z; }

// `.length()` of arrays and vectors (the translator writes length_of(name))
template <class T, int N> SF_HD constexpr int length_of(const T (&)[N]) { return N; }
SF_HD constexpr int length_of(const vec2&) { return 2; }
SF_HD constexpr int length_of(const vec3&) { return 3; }
SF_HD constexpr int length_of(const vec4&) { return 4; }

// GLSL int(x)/uint(x)/float(x)/bool(x) constructors (§5.4.1); NaN and out-of-range values convert by sfmath's rule
SF_HD int to_int(float x) { return sf::to_int(x); }
SF_HD int to_int(int x) { return x; }
SF_HD int to_int(uint x) { return (int)x; }
SF_HD int to_int(bool x) { return x ? 1 : 0; }
SF_HD uint to_uint(float x) { return (x != x || x <= 0.0f) ? 0u : ((x >= 4294967296.0f) ? 4294967295u : (uint)x); }
SF_HD uint to_uint(int x) { return (uint)x; }
SF_HD uint to_uint(uint x) { return x; }
SF_HD uint to_uint(bool x) { return x ? 1u : 0u; }
SF_HD ivec2::ivec2(const vec2& v) { x = sf::to_int(v.x); y = sf::to_int(v.y); }
SF_HD ivec3::ivec3(const vec3& v) { x = sf::to_int(v.x); y = sf::to_int(v.y); z = sf::to_int(v.z); }
SF_HD ivec4::ivec4(const vec4& v) { x = sf::to_int(v.x); y = sf::to_int(v.y); z = sf::to_int(v.z); w = sf::to_int(v.w); }
SF_HD uvec2::uvec2(const vec2& v) { x = to_uint(v.x); y = to_uint(v.y); }
SF_HD uvec3::uvec3(const vec3& v) { x = to_uint(v.x); y = to_uint(v.y); z = to_uint(v.z); }
SF_HD uvec4::uvec4(const vec4& v) { x = to_uint(v.x); y = to_uint(v.y); z = to_uint(v.z); w = to_uint(v.w); }
SF_HD ivec2::ivec2(const uvec2& v) { x = (int)v.x; y = (int)v.y; }
SF_HD ivec3::ivec3(const uvec3& v) { x = (int)v.x; y = (int)v.y; z = (int)v.z; }
SF_HD ivec4::ivec4(const uvec4& v) { x = (int)v.x; y = (int)v.y; z = (int)v.z; w = (int)v.w; }
SF_HD uvec2::uvec2(const ivec2& v) { x = (uint)v.x; y = (uint)v.y; }
SF_HD uvec3::uvec3(const ivec3& v) { x = (uint)v.x; y = (uint)v.y; z = (uint)v.z; }
SF_HD uvec4::uvec4(const ivec4& v) { x = (uint)v.x; y = (uint)v.y; z = (uint)v.z; w = (uint)v.w; }

#define SF_RT_INT_OP(V, T, N, op) \
    SF_HD V operator op(const V& a, const V& b) { V r; for (int k = 0; k < N; k++) r.d[k] = a.d[k] op b.d[k]; return r; } \
    SF_HD V operator op(const V& a, T s) { V r; for (int k = 0; k < N; k++) r.d[k] = a.d[k] op s; return r; } \
    SF_HD V operator op(T s, const V& a) { V r; for (int k = 0; k < N; k++) r.d[k] = s op a.d[k]; return r; } \
    SF_HD V& operator op##=(V& a, const V& b) { for (int k = 0; k < N; k++) a.d[k] = a.d[k] op b.d[k]; return a; } \
    SF_HD V& operator op##=(V& a, T s) { for (int k = 0; k < N; k++) a.d[k] = a.d[k] op s; return a; }
#define SF_RT_INT_OPS(V, T, N) \
    SF_RT_INT_OP(V, T, N, +) SF_RT_INT_OP(V, T, N, -) SF_RT_INT_OP(V, T, N, *) SF_RT_INT_OP(V, T, N, /) SF_RT_INT_OP(V, T, N, %) \
    SF_RT_INT_OP(V, T, N, &) SF_RT_INT_OP(V, T, N, |) SF_RT_INT_OP(V, T, N, ^) SF_RT_INT_OP(V, T, N, <<) SF_RT_INT_OP(V, T, N, >>) \
    SF_HD V operator-(const V& a) { V r; for (int k = 0; k < N; k++) r.d[k] = (T)(0 - a.d[k]); return r; } \
    SF_HD V operator~(const V& a) { V r; for (int k = 0; k < N; k++) r.d[k] = ~a.d[k]; return r; } \
    SF_HD bool operator==(const V& a, const V& b) { bool e = true; for (int k = 0; k < N; k++) e = e && (a.d[k] == b.d[k]); return e; } \
    SF_HD bool operator!=(const V& a, const V& b) { return !(a == b); }
SF_RT_INT_OPS(ivec2, int, 2) SF_RT_INT_OPS(ivec3, int, 3) SF_RT_INT_OPS(ivec4, int, 4)
SF_RT_INT_OPS(uvec2, uint, 2) SF_RT_INT_OPS(uvec3, uint, 3) SF_RT_INT_OPS(uvec4, uint, 4)

#define SF_RT_ARITH(V, N) \
    SF_HD V operator+(const V& a, const V& b) { V r; for (int k = 0; k < N; k++) r.d[k] = a.d[k] + b.d[k]; return r; } \
    SF_HD V operator-(const V& a, const V& b) { V r; for (int k = 0; k < N; k++) r.d[k] = a.d[k] - b.d[k]; return r; } \
    SF_HD V operator*(const V& a, const V& b) { V r; for (int k = 0; k < N; k++) r.d[k] = a.d[k]*b.d[k]; return r; } \
    SF_HD V operator/(const V& a, const V& b) { V r; for (int k = 0; k < N; k++) r.d[k] = a.d[k]/b.d[k]; return r; } \
    SF_HD V operator+(const V& a, float s) { V r; for (int k = 0; k < N; k++) r.d[k] = a.d[k] + s; return r; } \
    SF_HD V operator-(const V& a, float s) { V r; for (int k = 0; k < N; k++) r.d[k] = a.d[k] - s; return r; } \
    SF_HD V operator*(const V& a, float s) { V r; for (int k = 0; k < N; k++) r.d[k] = a.d[k]*s; return r; } \
    SF_HD V operator/(const V& a, float s) { V r; for (int k = 0; k < N; k++) r.d[k] = a.d[k]/s; return r; } \
    SF_HD V operator+(float s, const V& a) { V r; for (int k = 0; k < N; k++) r.d[k] = s + a.d[k]; return r; } \
    SF_HD V operator-(float s, const V& a) { V r; for (int k = 0; k < N; k++) r.d[k] = s - a.d[k]; return r; } \
    SF_HD V operator*(float s, const V& a) { V r; for (int k = 0; k < N; k++) r.d[k] = s*a.d[k]; return r; } \
    SF_HD V operator/(float s, const V& a) { V r; for (int k = 0; k < N; k++) r.d[k] = s/a.d[k]; return r; } \
    SF_HD V operator-(const V& a) { V r; for (int k = 0; k < N; k++) r.d[k] = -a.d[k]; return r; } \
    SF_HD V operator+(const V& a) { return a; } \
    SF_HD bool operator==(const V& a, const V& b) { bool e = true; for (int k = 0; k < N; k++) e = e && (a.d[k] == b.d[k]); return e; } \
    SF_HD bool operator!=(const V& a, const V& b) { return !(a == b); }
SF_RT_ARITH(vec2, 2)
SF_RT_ARITH(vec3, 3)
SF_RT_ARITH(vec4, 4)

// ---- built-in functions (GLSL 3.30 §8.1-8.3) --------------------------------------------------------------------------
// scalar forms; every transcendental is sfmath's fixed sequence of binary32 operations
SF_HD float radians(float x) { return x*(PI/180.0f); }
SF_HD float degrees(float x) { return x*(180.0f/PI); }
SF_HD float sin(float x) { return sf::sin(x); }
SF_HD float cos(float x) { return sf::cos(x); }
SF_HD float tan(float x) { return sf::sin(x)/sf::cos(x); }
SF_HD float atan(float x) { return sf::atan(x); }
SF_HD float atan(float y, float x) { return sf::atan(y, x); }
SF_HD float asin(float x) { return sf::atan(x, sf::sqrt((1.0f - x)*(1.0f + x))); }
SF_HD float acos(float x) { return sf::atan(sf::sqrt((1.0f - x)*(1.0f + x)), x); }
SF_HD float exp(float x) { return sf::exp(x); }
SF_HD float log(float x) { return sf::log(x); }
SF_HD float exp2(float x) { return sf::exp2(x); }
SF_HD float log2(float x) { return sf::log2(x); }
SF_HD float pow(float x, float y) { return sf::pow(x, y); }
SF_HD float sqrt(float x) { return sf::sqrt(x); }
SF_HD float inversesqrt(float x) { return 1.0f/sf::sqrt(x); }
SF_HD float sinh(float x) { return (sf::exp(x) - sf::exp(-x))*0.5f; }
SF_HD float cosh(float x) { return (sf::exp(x) + sf::exp(-x))*0.5f; }
SF_HD float tanh(float x) { const float e = sf::exp(-2.0f*sf::abs(x)); const float t = (1.0f - e)/(1.0f + e); return (x < 0.0f) ? -t : t; }
SF_HD float abs(float x) { return sf::abs(x); }
SF_HD float sign(float x) { return sf::sign(x); }
SF_HD float floor(float x) { return ::floorf(x); }
SF_HD float ceil(float x) { return ::ceilf(x); }
SF_HD float trunc(float x) { return ::truncf(x); }
SF_HD float round(float x) { return ::floorf(x + 0.5f); }
SF_HD float roundEven(float x) { return ::rintf(x); }
SF_HD float fract(float x) { return sf::fract(x); }
SF_HD float mod(float x, float y) { return sf::mod(x, y); }
SF_HD float min(float a, float b) { return sf::min(a, b); }
SF_HD float max(float a, float b) { return sf::max(a, b); }
SF_HD float clamp(float x, float lo, float hi) { return sf::clamp(x, lo, hi); }
SF_HD float mix(float a, float b, float t) { return sf::mix(a, b, t); }
SF_HD float step(float edge, float x) { return (x < edge) ? 0.0f : 1.0f; }
SF_HD float smoothstep(float e0, float e1, float x) { return sf::smoothstep(e0, e1, x); }
SF_HD float fma(float a, float b, float c) { return ::fmaf(a, b, c); }
SF_HD bool isnan(float x) { return x != x; }
SF_HD bool isinf(float x) { return sf::abs(x) == INFINITY; }
SF_HD float length(float x) { return sf::abs(x); }
SF_HD float distance(float a, float b) { return sf::abs(a - b); }
SF_HD float dot(float a, float b) { return a*b; }
SF_HD float normalize(float x) { return sf::sign(x); }
SF_HD int floatBitsToInt(float x) { return (int)f2u(x); }
SF_HD uint floatBitsToUint(float x) { return f2u(x); }
SF_HD float intBitsToFloat(int x) { return u2f((uint32_t)x); }
SF_HD float uintBitsToFloat(uint x) { return u2f(x); }

SF_HD ivec2 floatBitsToInt(const vec2& v) { return ivec2(floatBitsToInt(v.x), floatBitsToInt(v.y)); }
SF_HD ivec3 floatBitsToInt(const vec3& v) { return ivec3(floatBitsToInt(v.x), floatBitsToInt(v.y), floatBitsToInt(v.z)); }
SF_HD ivec4 floatBitsToInt(const vec4& v) { return ivec4(floatBitsToInt(v.x), floatBitsToInt(v.y), floatBitsToInt(v.z), floatBitsToInt(v.w)); }
SF_HD uvec2 floatBitsToUint(const vec2& v) { return uvec2(f2u(v.x), f2u(v.y)); }
SF_HD uvec3 floatBitsToUint(const vec3& v) { return uvec3(f2u(v.x), f2u(v.y), f2u(v.z)); }
SF_HD uvec4 floatBitsToUint(const vec4& v) { return uvec4(f2u(v.x), f2u(v.y), f2u(v.z), f2u(v.w)); }
SF_HD vec2 intBitsToFloat(const ivec2& v) { return vec2(intBitsToFloat(v.x), intBitsToFloat(v.y)); }
SF_HD vec3 intBitsToFloat(const ivec3& v) { return vec3(intBitsToFloat(v.x), intBitsToFloat(v.y), intBitsToFloat(v.z)); }
SF_HD vec4 intBitsToFloat(const ivec4& v) { return vec4(intBitsToFloat(v.x), intBitsToFloat(v.y), intBitsToFloat(v.z), intBitsToFloat(v.w)); }
SF_HD vec2 uintBitsToFloat(const uvec2& v) { return vec2(u2f(v.x), u2f(v.y)); }
SF_HD vec3 uintBitsToFloat(const uvec3& v) { return vec3(u2f(v.x), u2f(v.y), u2f(v.z)); }
SF_HD vec4 uintBitsToFloat(const uvec4& v) { return vec4(u2f(v.x), u2f(v.y), u2f(v.z), u2f(v.w)); }

// integer forms: exact-match templates, so that (float, int) arguments pick the float overloads like GLSL does
#define SF_RT_INTS(A, B) typename std::enable_if<std::is_integral<A>::value && std::is_integral<B>::value && !std::is_same<A, bool>::value, int>::type
template <class A, class B> SF_HD SF_RT_INTS(A, B) min(A a, B b) { return ((int)b < (int)a) ? (int)b : (int)a; }
template <class A, class B> SF_HD SF_RT_INTS(A, B) max(A a, B b) { return ((int)a < (int)b) ? (int)b : (int)a; }
template <class A, class B, class C> SF_HD typename std::enable_if<std::is_integral<A>::value && std::is_integral<B>::value && std::is_integral<C>::value, int>::type
clamp(A x, B lo, C hi) { const int v = ((int)x < (int)lo) ? (int)lo : (int)x; return (v > (int)hi) ? (int)hi : v; }
SF_HD int abs(int x) { return (x < 0) ? -x : x; }
SF_HD int sign(int x) { return (x > 0) - (x < 0); }

#define SF_RT_MAP1(name) \
    SF_HD vec2 name(const vec2& a) { return vec2(name(a.x), name(a.y)); } \
    SF_HD vec3 name(const vec3& a) { return vec3(name(a.x), name(a.y), name(a.z)); } \
    SF_HD vec4 name(const vec4& a) { return vec4(name(a.x), name(a.y), name(a.z), name(a.w)); }
SF_RT_MAP1(radians) SF_RT_MAP1(degrees) SF_RT_MAP1(sin) SF_RT_MAP1(cos) SF_RT_MAP1(tan) SF_RT_MAP1(asin) SF_RT_MAP1(acos)
SF_RT_MAP1(atan) SF_RT_MAP1(exp) SF_RT_MAP1(log) SF_RT_MAP1(exp2) SF_RT_MAP1(log2) SF_RT_MAP1(sqrt) SF_RT_MAP1(inversesqrt)
SF_RT_MAP1(sinh) SF_RT_MAP1(cosh) SF_RT_MAP1(tanh) SF_RT_MAP1(abs) SF_RT_MAP1(sign) SF_RT_MAP1(floor) SF_RT_MAP1(ceil)
SF_RT_MAP1(trunc) SF_RT_MAP1(round) SF_RT_MAP1(roundEven) SF_RT_MAP1(fract)

#define SF_RT_MAP2(name) \
    SF_HD vec2 name(const vec2& a, const vec2& b) { return vec2(name(a.x, b.x), name(a.y, b.y)); } \
    SF_HD vec3 name(const vec3& a, const vec3& b) { return vec3(name(a.x, b.x), name(a.y, b.y), name(a.z, b.z)); } \
    SF_HD vec4 name(const vec4& a, const vec4& b) { return vec4(name(a.x, b.x), name(a.y, b.y), name(a.z, b.z), name(a.w, b.w)); }
#define SF_RT_MAP2S(name) /* second argument scalar */ \
    SF_HD vec2 name(const vec2& a, float b) { return vec2(name(a.x, b), name(a.y, b)); } \
    SF_HD vec3 name(const vec3& a, float b) { return vec3(name(a.x, b), name(a.y, b), name(a.z, b)); } \
    SF_HD vec4 name(const vec4& a, float b) { return vec4(name(a.x, b), name(a.y, b), name(a.z, b), name(a.w, b)); }
SF_RT_MAP2(atan) SF_RT_MAP2(pow) SF_RT_MAP2(mod) SF_RT_MAP2(min) SF_RT_MAP2(max) SF_RT_MAP2(step)
SF_RT_MAP2S(mod) SF_RT_MAP2S(min) SF_RT_MAP2S(max)
SF_HD vec2 step(float e, const vec2& a) { return vec2(step(e, a.x), step(e, a.y)); }
SF_HD vec3 step(float e, const vec3& a) { return vec3(step(e, a.x), step(e, a.y), step(e, a.z)); }
SF_HD vec4 step(float e, const vec4& a) { return vec4(step(e, a.x), step(e, a.y), step(e, a.z), step(e, a.w)); }

#define SF_RT_MAP3(V, N) \
    SF_HD V clamp(const V& x, const V& lo, const V& hi) { V r; for (int k = 0; k < N; k++) r.d[k] = clamp(x.d[k], lo.d[k], hi.d[k]); return r; } \
    SF_HD V clamp(const V& x, float lo, float hi) { V r; for (int k = 0; k < N; k++) r.d[k] = clamp(x.d[k], lo, hi); return r; } \
    SF_HD V mix(const V& a, const V& b, const V& t) { V r; for (int k = 0; k < N; k++) r.d[k] = mix(a.d[k], b.d[k], t.d[k]); return r; } \
    SF_HD V mix(const V& a, const V& b, float t) { V r; for (int k = 0; k < N; k++) r.d[k] = mix(a.d[k], b.d[k], t); return r; } \
    SF_HD V smoothstep(const V& e0, const V& e1, const V& x) { V r; for (int k = 0; k < N; k++) r.d[k] = smoothstep(e0.d[k], e1.d[k], x.d[k]); return r; } \
    SF_HD V smoothstep(float e0, float e1, const V& x) { V r; for (int k = 0; k < N; k++) r.d[k] = smoothstep(e0, e1, x.d[k]); return r; } \
    SF_HD V fma(const V& a, const V& b, const V& c) { V r; for (int k = 0; k < N; k++) r.d[k] = ::fmaf(a.d[k], b.d[k], c.d[k]); return r; } \
    SF_HD float dot(const V& a, const V& b) { float s = a.d[0]*b.d[0]; for (int k = 1; k < N; k++) s = s + a.d[k]*b.d[k]; return s; } \
    SF_HD float length(const V& a) { return sf::sqrt(dot(a, a)); } \
    SF_HD float distance(const V& a, const V& b) { return length(a - b); } \
    SF_HD V normalize(const V& a) { return a/length(a); } \
    SF_HD V reflect(const V& i, const V& n) { return i - (2.0f*dot(n, i))*n; } \
    SF_HD V faceforward(const V& n, const V& i, const V& nref) { return (dot(nref, i) < 0.0f) ? n : -n; } \
    SF_HD V refract(const V& i, const V& n, float eta) { \
        const float c = dot(n, i); const float k = 1.0f - eta*eta*(1.0f - c*c); \
        if (k < 0.0f) return V(0.0f); \
        return eta*i - (eta*c + sf::sqrt(k))*n; }
SF_RT_MAP3(vec2, 2)
SF_RT_MAP3(vec3, 3)
SF_RT_MAP3(vec4, 4)
SF_HD vec3 cross(const vec3& a, const vec3& b) { return vec3(a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y); }

// screen-space derivatives (§8.8): differences inside the 2 x 2 quad the unfused kernel lays out for fragments that use them
// (render_kernels.hpp, QUADS) — the "fine" form: each row and column of the quad has its own difference. The host build used
// by the CPU tests has no neighbours and returns zero.
template <int CTRL> SF_HD float quad_value(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, true));
#else
    return v;
#endif
}
SF_HD float dFdx(float v) { return quad_value<0xF5>(v) - quad_value<0xA0>(v); }      // quad_perm (1,1,3,3) - (0,0,2,2)
SF_HD float dFdy(float v) { return quad_value<0xEE>(v) - quad_value<0x44>(v); }      // quad_perm (2,3,2,3) - (0,1,0,1)
SF_HD float fwidth(float v) { return sf::abs(dFdx(v)) + sf::abs(dFdy(v)); }
SF_RT_MAP1(dFdx) SF_RT_MAP1(dFdy) SF_RT_MAP1(fwidth)

// mix with a boolean selector (§8.3: components of b where the selector is true) and modf
template <class B, typename std::enable_if<std::is_same<B, bool>::value, int>::type = 0>
SF_HD float mix(float a, float b, B t) { return t ? b : a; }              // exactly bool: mix(a, b, 1) stays the float form
SF_HD vec2 mix(const vec2& a, const vec2& b, const bvec2& t) { return vec2(t.x ? b.x : a.x, t.y ? b.y : a.y); }
SF_HD vec3 mix(const vec3& a, const vec3& b, const bvec3& t) { return vec3(t.x ? b.x : a.x, t.y ? b.y : a.y, t.z ? b.z : a.z); }
SF_HD vec4 mix(const vec4& a, const vec4& b, const bvec4& t) { return vec4(t.x ? b.x : a.x, t.y ? b.y : a.y, t.z ? b.z : a.z, t.w ? b.w : a.w); }
SF_HD float modf(float x, float& whole) { whole = ::truncf(x); return x - whole; }
SF_HD vec2 modf(const vec2& x, vec2& whole) { whole = trunc(x); return x - whole; }
SF_HD vec3 modf(const vec3& x, vec3& whole) { whole = trunc(x); return x - whole; }
SF_HD vec4 modf(const vec4& x, vec4& whole) { whole = trunc(x); return x - whole; }

// vector relational functions (§8.6)
#define SF_RT_REL3(name, op, V2, V3, V4) \
    SF_HD bvec2 name(const V2& a, const V2& b) { return {a.x op b.x, a.y op b.y}; } \
    SF_HD bvec3 name(const V3& a, const V3& b) { return {a.x op b.x, a.y op b.y, a.z op b.z}; } \
    SF_HD bvec4 name(const V4& a, const V4& b) { return {a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w}; }
#define SF_RT_REL(name, op) SF_RT_REL3(name, op, vec2, vec3, vec4) SF_RT_REL3(name, op, ivec2, ivec3, ivec4) SF_RT_REL3(name, op, uvec2, uvec3, uvec4)
SF_RT_REL(lessThan, <) SF_RT_REL(lessThanEqual, <=) SF_RT_REL(greaterThan, >) SF_RT_REL(greaterThanEqual, >=)
SF_RT_REL(equal, ==) SF_RT_REL(notEqual, !=)
SF_HD bool any(bvec2 b) { return b.x || b.y; }
SF_HD bool any(bvec3 b) { return b.x || b.y || b.z; }
SF_HD bool any(bvec4 b) { return b.x || b.y || b.z || b.w; }
SF_HD bool all(bvec2 b) { return b.x && b.y; }
SF_HD bool all(bvec3 b) { return b.x && b.y && b.z; }
SF_HD bool all(bvec4 b) { return b.x && b.y && b.z && b.w; }
SF_HD bvec2 not_(bvec2 b) { return {!b.x, !b.y}; }
SF_HD bvec3 not_(bvec3 b) { return {!b.x, !b.y, !b.z}; }
SF_HD bvec4 not_(bvec4 b) { return {!b.x, !b.y, !b.z, !b.w}; }

// ---- matrices (column major, §5.4.2, §5.9-5.10) -----------------------------------------------------------------------
template <class V, int N>
struct matn {
    V c[N];
    SF_HD matn() { for (int j = 0; j < N; j++) c[j] = V(0.0f); }
    SF_HD explicit matn(float s) { for (int j = 0; j < N; j++) { c[j] = V(0.0f); c[j].d[j] = s; } }
    template <class... A, typename std::enable_if<sizeof...(A) == N*N && (N > 1), int>::type = 0>
    SF_HD matn(A... values) { const float v[] = {(float)values...}; for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) c[j].d[i] = v[j*N + i]; }
    template <class... A, typename std::enable_if<sizeof...(A) == N && (std::is_convertible<A, V>::value && ...) && !(std::is_arithmetic<A>::value || ...), int>::type = 0>
    SF_HD matn(const A&... columns) { const V v[] = {V(columns)...}; for (int j = 0; j < N; j++) c[j] = v[j]; }
    // from a matrix of another size (§5.4.2): the common block is copied, the rest is the identity
    template <class W, int M, typename std::enable_if<M != N, int>::type = 0>
    SF_HD explicit matn(const matn<W, M>& o) {
        for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) c[j].d[i] = (i < M && j < M) ? o.c[j].d[i] : ((i == j) ? 1.0f : 0.0f);
    }
    SF_HD V& operator[](int j) { return c[j]; }
    SF_HD const V& operator[](int j) const { return c[j]; }
    SF_HD matn& operator*=(const matn& o) { *this = *this*o; return *this; }
    SF_HD matn& operator*=(float s) { for (int j = 0; j < N; j++) c[j] *= s; return *this; }
};
typedef matn<vec2, 2> mat2; typedef matn<vec3, 3> mat3; typedef matn<vec4, 4> mat4;
typedef mat2 mat2x2; typedef mat3 mat3x3; typedef mat4 mat4x4;

template <class V, int N> SF_HD V operator*(const matn<V, N>& m, const V& v) {       // (m*v)[i] = sum_j m[j][i]*v[j], left to right
    V r;
    for (int i = 0; i < N; i++) { float s = m.c[0].d[i]*v.d[0]; for (int j = 1; j < N; j++) s = s + m.c[j].d[i]*v.d[j]; r.d[i] = s; }
    return r;
}
template <class V, int N> SF_HD V operator*(const V& v, const matn<V, N>& m) {       // (v*m)[j] = dot(v, m[j])
    V r;
    for (int j = 0; j < N; j++) r.d[j] = dot(v, m.c[j]);
    return r;
}
template <class V, int N> SF_HD matn<V, N> operator*(const matn<V, N>& a, const matn<V, N>& b) { matn<V, N> r; for (int j = 0; j < N; j++) r.c[j] = a*b.c[j]; return r; }
template <class V, int N> SF_HD matn<V, N> operator*(const matn<V, N>& a, float s) { matn<V, N> r; for (int j = 0; j < N; j++) r.c[j] = a.c[j]*s; return r; }
template <class V, int N> SF_HD matn<V, N> operator*(float s, const matn<V, N>& a) { matn<V, N> r; for (int j = 0; j < N; j++) r.c[j] = s*a.c[j]; return r; }
template <class V, int N> SF_HD matn<V, N> operator/(const matn<V, N>& a, float s) { matn<V, N> r; for (int j = 0; j < N; j++) r.c[j] = a.c[j]/s; return r; }
template <class V, int N> SF_HD matn<V, N> operator+(const matn<V, N>& a, const matn<V, N>& b) { matn<V, N> r; for (int j = 0; j < N; j++) r.c[j] = a.c[j] + b.c[j]; return r; }
template <class V, int N> SF_HD matn<V, N> operator-(const matn<V, N>& a, const matn<V, N>& b) { matn<V, N> r; for (int j = 0; j < N; j++) r.c[j] = a.c[j] - b.c[j]; return r; }
template <class V, int N> SF_HD matn<V, N> operator-(const matn<V, N>& a) { matn<V, N> r; for (int j = 0; j < N; j++) r.c[j] = -a.c[j]; return r; }
template <class V, int N> SF_HD matn<V, N> matrixCompMult(const matn<V, N>& a, const matn<V, N>& b) { matn<V, N> r; for (int j = 0; j < N; j++) r.c[j] = a.c[j]*b.c[j]; return r; }
template <class V, int N> SF_HD matn<V, N> transpose(const matn<V, N>& a) { matn<V, N> r; for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) r.c[j].d[i] = a.c[i].d[j]; return r; }
// a swizzle on either side of a matrix product converts like any other argument
SF_HD vec2 operator*(const mat2& m, const vec2& v) { return operator*<vec2, 2>(m, v); }
SF_HD vec3 operator*(const mat3& m, const vec3& v) { return operator*<vec3, 3>(m, v); }
SF_HD vec4 operator*(const mat4& m, const vec4& v) { return operator*<vec4, 4>(m, v); }
SF_HD vec2 operator*(const vec2& v, const mat2& m) { return operator*<vec2, 2>(v, m); }
SF_HD vec3 operator*(const vec3& v, const mat3& m) { return operator*<vec3, 3>(v, m); }
SF_HD vec4 operator*(const vec4& v, const mat4& m) { return operator*<vec4, 4>(v, m); }
SF_HD vec2& operator*=(vec2& v, const mat2& m) { v = v*m; return v; }
SF_HD vec3& operator*=(vec3& v, const mat3& m) { v = v*m; return v; }
SF_HD vec4& operator*=(vec4& v, const mat4& m) { v = v*m; return v; }
SF_HD mat2 outerProduct(const vec2& c, const vec2& r) { return mat2(c*r.x, c*r.y); }
SF_HD mat3 outerProduct(const vec3& c, const vec3& r) { return mat3(c*r.x, c*r.y, c*r.z); }

SF_HD float determinant(const mat2& m) { return m.c[0].x*m.c[1].y - m.c[1].x*m.c[0].y; }
SF_HD float determinant(const mat3& m) {
    return m.c[0].x*(m.c[1].y*m.c[2].z - m.c[2].y*m.c[1].z) - m.c[1].x*(m.c[0].y*m.c[2].z - m.c[2].y*m.c[0].z) + m.c[2].x*(m.c[0].y*m.c[1].z - m.c[1].y*m.c[0].z);
}
SF_HD mat2 inverse(const mat2& m) { const float d = determinant(m); return mat2(m.c[1].y/d, -m.c[0].y/d, -m.c[1].x/d, m.c[0].x/d); }
SF_HD mat3 inverse(const mat3& m) {
    const vec3 a = m.c[0], b = m.c[1], c = m.c[2];
    const vec3 r0 = cross(b, c), r1 = cross(c, a), r2 = cross(a, b);
    const float d = dot(a, r0);
    return transpose(mat3(r0/d, r1/d, r2/d));
}
SF_HD float minor3(const mat4& m, int skip_column, int skip_row) {
    float v[9]; int n = 0;
    for (int j = 0; j < 4; j++) if (j != skip_column) for (int i = 0; i < 4; i++) if (i != skip_row) v[n++] = m.c[j].d[i];
    return v[0]*(v[4]*v[8] - v[7]*v[5]) - v[3]*(v[1]*v[8] - v[7]*v[2]) + v[6]*(v[1]*v[5] - v[4]*v[2]);
}
SF_HD float determinant(const mat4& m) {
    float d = 0.0f;
    for (int j = 0; j < 4; j++) { const float t = m.c[j].d[0]*minor3(m, j, 0); d = (j & 1) ? d - t : d + t; }
    return d;
}
SF_HD mat4 inverse(const mat4& m) {
    const float d = determinant(m);
    mat4 r;
    for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) { const float c = minor3(m, j, i); r.c[i].d[j] = (((i + j) & 1) ? -c : c)/d; }
    return r;
}

// ---- samplers (§8.7) --------------------------------------------------------------------------------------------------
// One sampler of a translated fragment may be served from an LDS tile (SF_JIT_TILE_SLOT, chosen by the translator for tap-heavy
// fragments — blurs, feedback kernels). The tile is a cache of decoded texels over a rectangle of UNWRAPPED texel indices,
// entry (x, y) = texel(t, wrap(x), wrap(y)): a tap whose footprint lies inside reads LDS, any other tap takes the generic
// fetch, and both blend the same floats with the same operations — results never depend on what was staged. The rectangle
// comes from running the fragment once at the block's first and last sample in `record` mode (JitShader::setup below).
#ifndef SF_JIT_TILE_SLOT
#define SF_JIT_TILE_SLOT -1
#endif
constexpr int TILE_TEXELS = 3072;                    // 48 KiB of float4
struct TileView {
    const float4* texels;                            // nullptr: this sampler has no tile
    int* record;                                     // not null: note the texels the taps touch {lo x, lo y, hi x, hi y} instead of reading the tile
    int x0, y0, w;                                   // staged rectangle [x0, x0 + w) x [y0, y0 + h), row pitch w
    unsigned nearest_w, nearest_h, linear_w, linear_h;   // w, h and max(w - 1, 0), max(h - 1, 0): the bounds of a tap's first texel
};
// (the view travels by value inside the sampler: a pointer from the fragment object to one of its own members would keep the
// whole object out of registers)
struct sampler2D { const Tex* t; TileView tile; };
SF_HD vec4 up(sf::vec4 c) { return vec4(c.x, c.y, c.z, c.w); }
SF_HD vec3 up(sf::vec3 c) { return vec3(c.x, c.y, c.z); }
SF_HD vec2 up(sf::vec2 c) { return vec2(c.x, c.y); }
SF_HD sf::vec2 lo(const vec2& c) { return {c.x, c.y}; }
SF_HD sf::vec3 lo(const vec3& c) { return {c.x, c.y, c.z}; }
// an unbound sampler reads as opaque black texels, like an incomplete GL texture
SF_HD int tile_bound(float texel) { return (int)sf::min(sf::max(texel, -16777216.0f), 16777216.0f); }
SF_HD sf::vec4 tiled_texture(const Tex& t, const TileView& v, sf::vec2 uv) {      // sf::texture (glsl.hpp) with the texel fetch swapped
    if (t.filter >= FILTER_LINEAR_MIPMAP) return sf::texture(t, uv);       // mipmapped: the tile holds level 0 only
    const float u = uv.x*(float)t.width, w = uv.y*(float)t.height;
    const bool nearest = t.filter == FILTER_NEAREST;
    const float ub = nearest ? u : u - 0.5f, vb = nearest ? w : w - 0.5f;
    const float fu = ::floorf(ub), fv = ::floorf(vb);
    if (v.record) {
        const int i = tile_bound(fu), j = tile_bound(fv), far = nearest ? 0 : 1;
        int* box = v.record;
        if (i < box[0]) box[0] = i;
        if (j < box[1]) box[1] = j;
        if (i + far > box[2]) box[2] = i + far;
        if (j + far > box[3]) box[3] = j + far;
        return sf::texture(t, uv);
    }
    const unsigned rx = (unsigned)(int)fu - (unsigned)v.x0, ry = (unsigned)(int)fv - (unsigned)v.y0;
    if (nearest) {
        if (rx < v.nearest_w && ry < v.nearest_h) { const float4 c = v.texels[ry*(unsigned)v.w + rx]; return {c.x, c.y, c.z, c.w}; }
        return sf::texture(t, uv);
    }
    if (rx < v.linear_w && ry < v.linear_h) {
        const float4* p = v.texels + (ry*(unsigned)v.w + rx);
        const float4 t00 = p[0], t10 = p[1], t01 = p[v.w], t11 = p[v.w + 1];
        const float a = ub - fu, b = vb - fv;
        const float na = 1.0f - a, nb = 1.0f - b;
        const float w00 = na*nb, w10 = a*nb, w01 = na*b, w11 = a*b;
        return {bilerp(w00, w10, w01, w11, t00.x, t10.x, t01.x, t11.x),
                bilerp(w00, w10, w01, w11, t00.y, t10.y, t01.y, t11.y),
                bilerp(w00, w10, w01, w11, t00.z, t10.z, t01.z, t11.z),
                bilerp(w00, w10, w01, w11, t00.w, t10.w, t01.w, t11.w)};
    }
    return sf::texture(t, uv);
}
SF_HD vec4 texture(sampler2D s, const vec2& uv) {
    if (!(s.t && s.t->data)) return vec4(0.0f, 0.0f, 0.0f, 1.0f);
    if (s.tile.texels) return up(tiled_texture(*s.t, s.tile, lo(uv)));
    return up(sf::texture(*s.t, lo(uv)));
}
SF_HD vec4 texture(sampler2D s, const vec2& uv, float) { return texture(s, uv); }          // one level: the bias selects nothing
SF_HD vec4 textureLod(sampler2D s, const vec2& uv, float) { return texture(s, uv); }
SF_HD vec4 texelFetch(sampler2D s, ivec2 p, int) { return (s.t && s.t->data) ? up(sf::texel_fetch(*s.t, p.x, p.y)) : vec4(0.0f, 0.0f, 0.0f, 1.0f); }
SF_HD ivec2 textureSize(sampler2D s, int) { return (s.t && s.t->data) ? ivec2(s.t->width, s.t->height) : ivec2(1, 1); }

// ---- prelude: include/shaderflow.glsl ---------------------------------------------------------------------------------
constexpr float SQRT2 = 1.4142135623730951f, SQRT3 = 1.7320508075688772f, SQRT5 = 2.2360679774997898f;     // :7-11 (PI, TAU: sfmath.hpp)
using sf::PI; using sf::TAU;

SF_HD float proportion(float a, float b, float c) { return (b*c)/a; }                                       // :24-26
SF_HD float lerp(float ax, float ay, float bx, float by, float x) { return ay + (x - ax)*(by - ay)/(bx - ax); }   // :29-31
SF_HD float smoothlerp(float a, float b, float difference) {                                                // :37-41
    const float t = clamp((a - b)/difference + 0.5f, 0.0f, 1.0f);
    const float offset = difference*t*(1.0f - t)/2.0f;
    return mix(a, b, t) - offset;
}
SF_HD float smin(float a, float b, float k) { return smoothlerp(a, b, k); }                                 // :44-47
SF_HD float smax(float a, float b, float k) { return smoothlerp(a, b, -k); }
SF_HD float smin(float a, float b) { return smoothlerp(a, b, 1.0f); }
SF_HD float smax(float a, float b) { return smoothlerp(a, b, -1.0f); }
SF_HD float smoothmix(float a, float b, float x0, float x1, float x) { return mix(a, b, smoothstep(x0, x1, x)); }   // :51-53
SF_HD float smix(float a, float b, float x0, float x1, float x) { return smoothmix(a, b, x0, x1, x); }
SF_HD float triangle_wave(float x, float period) { return 2.0f*abs(mod(2.0f*x/period - 0.5f, 2.0f) - 1.0f) - 1.0f; }   // :63-66
SF_HD float angle(const vec4& a, const vec4& b) { return acos(dot(a, b)/(length(a)*length(b))); }           // :71-73
SF_HD float angle(const vec3& a, const vec3& b) { return acos(dot(a, b)/(length(a)*length(b))); }
SF_HD float angle(const vec2& a, const vec2& b) { return acos(dot(a, b)/(length(a)*length(b))); }
SF_HD mat2 rotate2d(float a) { return mat2(cos(a), -sin(a), sin(a), cos(a)); }                              // :75-77
#define rotate2deg(a) rotate2d(radians(a))
SF_HD vec3 rotate3d(const vec3& v, const vec3& axis, float a) { return mix(dot(axis, v)*axis, v, cos(a)) + cross(axis, v)*sin(a); }   // :81-83
#define rotate3deg(v, axis, a) rotate3d(v, axis, radians(a))
SF_HD vec2 stuv2gluv(const vec2& stuv) { return (stuv*2.0f) - 1.0f; }                                       // :89-96
SF_HD vec2 s2g(const vec2& stuv) { return stuv2gluv(stuv); }
SF_HD vec2 gluv2stuv(const vec2& gluv) { return (gluv + 1.0f)/2.0f; }
SF_HD vec2 g2s(const vec2& gluv) { return gluv2stuv(gluv); }
SF_HD vec2 stuv2stxy(const vec2& stuv, const vec2& resolution) { return resolution*stuv; }                  // :103
SF_HD vec2 stxy2stuv(const vec2& stxy, const vec2& resolution) { return stxy/resolution; }                  // :107
SF_HD vec2 agluv_mirrored_repeat(const vec2& agluv) { return vec2(triangle_wave(agluv.x, 4.0f), triangle_wave(agluv.y, 4.0f)); }   // :119-124
SF_HD bool astuv_oob(const vec2& a) { return (a.x < 0.0f) || (a.x > 1.0f) || (a.y < 0.0f) || (a.y > 1.0f); }   // :135-137
SF_HD bool agluv_oob(const vec2& a) { return (a.x < -1.0f) || (a.x > 1.0f) || (a.y < -1.0f) || (a.y > 1.0f); }   // :141-143
SF_HD vec2 polar2rect(float radius, float a) { return radius*vec2(cos(a), sin(a)); }                        // :149-151
SF_HD vec3 sphere2rect(float radius, float theta, float phi) {                                              // :154-160
    return vec3(radius*sin(theta)*cos(phi), radius*sin(theta)*sin(phi), radius*cos(theta));
}
SF_HD vec4 gtexture(sampler2D image, const vec2& gluv) {                                                    // :165-169
    const vec2 resolution = textureSize(image, 0);
    const vec2 scale = vec2(resolution.y/resolution.x, 1.0f);
    return texture(image, gluv2stuv(gluv*scale));
}
SF_HD vec4 stexture(sampler2D image, const vec2& stuv) { return gtexture(image, stuv2gluv(stuv)); }         // :198-200
SF_HD vec4 astexture(sampler2D image, const vec2& astuv) { return texture(image, astuv); }                  // :202-204
SF_HD vec3 palette(float t, const vec3& A, const vec3& B, const vec3& C, const vec3& D) {                   // :208-216
    if (t < 0.25f) return mix(A, B, t*4.0f);
    if (t < 0.5f) return mix(B, C, (t - 0.25f)*4.0f);
    return mix(C, D, (t - 0.5f)*4.0f);
}
#define PALETTE_MAGMA_1 vec3(0.01060815f, 0.01808215f, 0.10018654f)
#define PALETTE_MAGMA_2 vec3(0.38092887f, 0.12061482f, 0.32506528f)
#define PALETTE_MAGMA_3 vec3(0.79650140f, 0.10506637f, 0.31063031f)
#define PALETTE_MAGMA_4 vec3(0.95922872f, 0.53307513f, 0.37488950f)
#define palette_magma(x) palette(x, PALETTE_MAGMA_1, PALETTE_MAGMA_2, PALETTE_MAGMA_3, PALETTE_MAGMA_4)
SF_HD bool isBlackKey(int index) { const int key = index % 12; return key == 1 || key == 3 || key == 6 || key == 8 || key == 10; }   // :227-244
SF_HD bool isBlackKey(float key) { return isBlackKey(sf::to_int(key)); }
SF_HD bool isWhiteKey(int index) { return !isBlackKey(index); }
SF_HD bool isWhiteKey(float key) { return isWhiteKey(sf::to_int(key)); }
SF_HD float _sdLine(const vec3& origin, const vec3& A, const vec3& B, bool segment) {                       // :255-261
    const vec3 direction = B - A, shortest = origin - A;
    float t = dot(shortest, direction)/dot(direction, direction);
    if (segment) t = clamp(t, 0.0f, 1.0f);
    return length(shortest - direction*t);
}
SF_HD float sdLine(const vec2& o, const vec2& p1, const vec2& p2) { return _sdLine(vec3(o, 0.0f), vec3(p1, 0.0f), vec3(p2, 0.0f), false); }   // :263-271
SF_HD float sdLine(const vec3& o, const vec3& p1, const vec3& p2) { return _sdLine(o, p1, p2, false); }
SF_HD float sdLineSegment(const vec3& o, const vec3& p1, const vec3& p2) { return _sdLine(o, p1, p2, true); }
SF_HD float sdLineSegment(const vec2& o, const vec2& p1, const vec2& p2) { return _sdLine(vec3(o, 0.0f), vec3(p1, 0.0f), vec3(p2, 0.0f), true); }
SF_HD float sdSphere(const vec3& origin, const vec3& position, float radius) { return length(position - origin) - radius; }   // :275-277
SF_HD float sdPlane(const vec3& origin, const vec3& point, const vec3& normal) { return dot(origin - point, normalize(normal)); }   // :280-282
SF_HD float sdBox(const vec3& origin, const vec3& point, const vec3& size) {                                // :285-288
    const vec3 d = abs(origin - point) - size/2.0f;
    return min(max(d.x, max(d.y, d.z)), 0.0f) + length(max(d, 0.0f));
}
SF_HD float sdOctahedron(const vec3& origin, const vec3& point, float size) { const vec3 p = abs(origin - point); return SQRT3*(p.x + p.y + p.z - size); }   // :291-294
SF_HD float sdUnion(float a, float b) { return min(a, b); }                                                 // :299-301
SF_HD float sdSmoothUnion(float a, float b, float width) { const float k = clamp(0.5f + 0.5f*(b - a)/width, 0.0f, 1.0f); return mix(b, a, k) - width*k*(1.0f - k); }
SF_HD float sdSubtraction(float a, float b) { return max(b, -a); }
SF_HD float sdSmoothSubtraction(float a, float b, float width) { const float k = clamp(0.5f - 0.5f*(b + a)/width, 0.0f, 1.0f); return mix(b, -a, k) + width*k*(1.0f - k); }
SF_HD float sdIntersection(float a, float b) { return max(a, b); }
SF_HD float sdSmoothIntersection(float a, float b, float width) { const float k = clamp(0.5f - 0.5f*(b - a)/width, 0.0f, 1.0f); return mix(b, a, k) + width*k*(1.0f - k); }
SF_HD vec4 blend(const vec4& a, const vec4& b) { return mix(a, b, b.a); }                                   // :345-347
SF_HD vec4 alpha_composite(const vec4& a, const vec4& b) { return a*(1.0f - b.a) + (b*b.a); }               // :351-353
SF_HD vec4 saturate(const vec4& c, float amount) { return clamp(c*amount, 0.0f, 1.0f); }                    // :356-358
SF_HD vec3 saturate(const vec3& c, float amount) { return clamp(c*amount, 0.0f, 1.0f); }
SF_HD vec2 saturate(const vec2& c, float amount) { return clamp(c*amount, 0.0f, 1.0f); }
SF_HD vec2 zoom(const vec2& uv, float z, const vec2& anchor) { return (uv - anchor)*(z*z) + anchor; }       // :363-369
SF_HD vec2 zoom(const vec2& uv, float z) { return uv*(z*z); }
SF_HD float atan_normalized(float x) { return 2.0f*atan(x)/PI; }                                            // :372-374
SF_HD float atan1(const vec2& p) { return atan(p.y, p.x); }
SF_HD float atan1n(const vec2& p) { return atan(p.y, p.x)/PI; }
SF_HD float atan2(float y, float x) { return (y < 0.0f) ? TAU - atan(-y, x) : atan(y, x); }                 // :384-390
SF_HD float atan2(const vec2& p) { return atan2(p.y, p.x); }
SF_HD float atan2n(float y, float x) { return atan2(y, x)/TAU; }
SF_HD float atan2n(const vec2& p) { return atan2n(p.y, p.x); }
SF_HD vec3 hsv2rgb(const vec3& hsv) { return up(sf::hsv2rgb(hsv.x, hsv.y, hsv.z)); }                       // :408-427 (glsl.hpp)
SF_HD vec3 hsv2rgb(float h, float s, float v) { return hsv2rgb(vec3(h, s, v)); }
SF_HD vec4 hsv2rgb(const vec4& hsv) { return vec4(hsv2rgb(vec3(hsv.x, hsv.y, hsv.z)), hsv.a); }
SF_HD vec3 rgb2hsv(const vec3& rgb) {                                                                       // :434-452
    const float cmax = max(rgb.r, max(rgb.g, rgb.b)), cmin = min(rgb.r, min(rgb.g, rgb.b));
    const float delta = cmax - cmin;
    float h = 0.0f;
    if (delta == 0.0f) h = 0.0f;
    else if (cmax == rgb.r) h = mod((rgb.g - rgb.b)/delta, 6.0f);
    else if (cmax == rgb.g) h = (rgb.b - rgb.r)/delta + 2.0f;
    else h = (rgb.r - rgb.g)/delta + 4.0f;
    h *= PI/3.0f;
    const float s = (cmax == 0.0f) ? 0.0f : delta/cmax;
    return vec3(h, s, cmax);
}
SF_HD vec3 rgb2hsv(float r, float g, float b) { return rgb2hsv(vec3(r, g, b)); }
SF_HD vec4 rgb2hsv(const vec4& rgb) { return vec4(rgb2hsv(vec3(rgb.x, rgb.y, rgb.z)), rgb.a); }
SF_HD float noise21(const vec2& p) { return fract(sin(dot(p, vec2(18.4835183f, 59.583596f)))*39758.381532f); }   // :460-471
SF_HD vec2 noise22(const vec2& p) { const float x = noise21(p); return vec2(x, noise21(p + x)); }
SF_HD float noise11(float f) { return fract(sin(f)*39758.381532f); }

// ---- prelude: include/complex.glsl:1-63 -------------------------------------------------------------------------------
SF_HD vec2 cadd(const vec2& a, const vec2& b) { return a + b; }
SF_HD vec2 csub(const vec2& a, const vec2& b) { return a - b; }
SF_HD float cmag(const vec2& a) { return length(a); }
SF_HD vec2 cpol(const vec2& a) { return vec2(length(a), atan(a.y, a.x)); }
SF_HD vec2 ccar(const vec2& p) { return vec2(p.x*cos(p.y), p.x*sin(p.y)); }
SF_HD vec2 cmul(const vec2& a, const vec2& b) { return vec2((a.x*b.x) - (a.y*b.y), (a.x*b.y) + (a.y*b.x)); }
SF_HD vec2 cdiv(const vec2& a, const vec2& b) { const float den = (b.x*b.x) + (b.y*b.y); return vec2(((a.x*b.x) + (a.y*b.y))/den, ((a.y*b.x) - (a.x*b.y))/den); }
SF_HD vec2 cconj(const vec2& a) { return vec2(a.x, -a.y); }
SF_HD vec2 cexp(const vec2& a) { const float e = exp(a.x); return vec2(e*cos(a.y), e*sin(a.y)); }

// ---- prelude: include/camera.glsl -------------------------------------------------------------------------------------
constexpr int CameraModeFreeCamera = 0, CameraMode2D = 1, CameraModeSpherical = 2;                          // :4-12
constexpr int CameraProjectionPerspective = 0, CameraProjectionStereoscopic = 1, CameraProjectionEquirectangular = 2;
struct Camera {                                                                                             // :14-52
    int mode, projection;
    vec3 position, up, down, left, right, forward, backward, zenith, origin, target;
    float orbital, dolly;
    vec3 plane_point, plane_normal;
    vec2 gluv, agluv, stuv, astuv, glxy, stxy;
    bool out_of_bounds;
    float separation, focal_length, isometric, zoom;
};
SF_HD vec3 CameraRectangle(const Camera& c, const vec2& gluv, float size) { return size*(gluv.x*c.right + gluv.y*c.up); }   // :55-57
SF_HD vec3 CameraRayOrigin(const Camera& c, const vec2& gluv) {                                             // :59-64
    return c.position + CameraRectangle(c, gluv, c.zoom*c.isometric) + (c.backward*c.orbital) + (c.backward*c.dolly);
}
SF_HD vec3 CameraRayTarget(const Camera& c, const vec2& gluv) {                                             // :66-71
    return c.position + CameraRectangle(c, gluv, c.zoom) + (c.backward*c.orbital) + (c.forward*c.focal_length);
}
#define GetCamera(name) Camera name = this->get_camera_()                                                   // :132-155

// ---- the fragment's environment ---------------------------------------------------------------------------------------
// Members a translated fragment reads by name: the varyings of vertex/default.glsl:1-17, the uniforms every scene
// sends (scene.py:687-703, camera.py:196-201, audio/module.py:413-421, spectrogram.py:313-320, waveform.py:89-90)
// and the macros of shaderflow.glsl:13-19.
struct FragmentBase {
    const Frag* frag_;
    vec2 fragCoord, stxy, glxy, stuv, astuv, gluv, agluv;
    vec4 gl_FragCoord, fragColor;
    int instance;
    bool discarded_;
    TileView tile_;                                  // set before load_user_() by a kernel that stages a tile; texels == nullptr: none
    float iTime, iTau, iDuration, iFrametime, iDeltatime, iCycle, iAspectRatio, iWidth, iHeight;
    vec2 iResolution, iMouse;
    float iWantAspect, iQuality, iSSAA, iFramerate;
    int iFrame, iLayer, iSubsample;
    bool iRealtime, iRendering, iMouseInside, iMouse1, iMouse2;
    int iCameraMode, iCameraProjection;
    vec3 iCameraRight, iCameraUpward, iCameraForward, iCameraPosition, iCameraZenith;
    float iCameraSeparation, iCameraZoom, iCameraIsometric, iCameraFocalLength, iCameraOrbital, iCameraDolly;
    float iAudioVolume, iAudioVolumeIntegral, iAudioSTD;
    int iSpectrogramLength, iSpectrogramBins, iSpectrogramScroll, iWaveformLength;
    bool iSpectrogramSmooth;
    float iSpectrogramOffset, iSpectrogramMin, iSpectrogramMax;

    SF_HD static vec3 v3(const float* p) { return vec3(p[0], p[1], p[2]); }
    SF_HD void load_(const Frag& f) {
        const Uniforms& u = *f.u;
        frag_ = &f;
        fragCoord = up(f.fragCoord); stxy = up(f.stxy); glxy = up(f.glxy); stuv = up(f.stuv); astuv = up(f.astuv); gluv = up(f.gluv); agluv = up(f.agluv);
        gl_FragCoord = vec4(f.fragCoord.x - 1.0f, f.fragCoord.y - 1.0f, 0.5f, 1.0f);     // stxy = iResolution*astuv + 1 (vertex/default.glsl:13)
        fragColor = vec4(0.0f); instance = 0; discarded_ = false; tile_ = TileView{};
        iTime = u.iTime; iTau = u.iTau; iDuration = u.iDuration;
        iFramerate = u.iFramerate; iFrametime = 1.0f/u.iFramerate; iDeltatime = iFrametime;    // shaderflow.glsl:13-14: the macro shadows the uniform
        iCycle = 2.0f*PI*u.iTau;                                                               // :15
        iResolution = vec2(u.iResolution[0], u.iResolution[1]);
        iAspectRatio = f.aspect; iWidth = iResolution.x; iHeight = iResolution.y;              // :16-18
        iMouse = vec2(u.iMouse[0], u.iMouse[1]);
        iWantAspect = u.iWantAspect; iQuality = u.iQuality; iSSAA = u.iSSAA;
        iFrame = u.iFrame; iLayer = u.iLayer; iSubsample = u.iSubsample;
        iRealtime = u.iRealtime != 0; iRendering = !iRealtime;                                 // :19
        iMouseInside = u.iMouseInside != 0; iMouse1 = u.iMouse1 != 0; iMouse2 = u.iMouse2 != 0;
        iCameraMode = u.iCameraMode; iCameraProjection = u.iCameraProjection;
        iCameraRight = v3(u.iCameraRight); iCameraUpward = v3(u.iCameraUpward); iCameraForward = v3(u.iCameraForward);
        iCameraPosition = v3(u.iCameraPosition); iCameraZenith = v3(u.iCameraZenith);
        iCameraSeparation = u.iCameraSeparation; iCameraZoom = u.iCameraZoom; iCameraIsometric = u.iCameraIsometric;
        iCameraFocalLength = u.iCameraFocalLength; iCameraOrbital = u.iCameraOrbital; iCameraDolly = u.iCameraDolly;
        iAudioVolume = u.iAudioVolume; iAudioVolumeIntegral = u.iAudioVolumeIntegral; iAudioSTD = u.iAudioSTD;
        iSpectrogramLength = u.iSpectrogramLength; iSpectrogramBins = u.iSpectrogramBins; iSpectrogramScroll = u.iSpectrogramScroll;
        iSpectrogramSmooth = u.iSpectrogramSmooth != 0; iWaveformLength = u.iWaveformLength;
        iSpectrogramOffset = u.iSpectrogramOffset; iSpectrogramMin = u.iSpectrogramMin; iSpectrogramMax = u.iSpectrogramMax;
    }
    // sampler slot k of the launch: the named slots are the per-frame copy (tape mode patches the audio textures), the rest is read in place
    SF_HD const Tex* texture_(int slot) const { return (slot < TEX_HISTORY) ? &frag_->tex[slot] : &frag_->history[slot - TEX_HISTORY]; }
    SF_HD sampler2D sampler_(int slot) const { return {texture_(slot), (slot == SF_JIT_TILE_SLOT) ? tile_ : TileView{}}; }
    SF_HD float user_(int slot) const { return frag_->u->user[slot]; }
    SF_HD int user_int_(int slot) const { return (int)f2u(frag_->u->user[slot]); }

    // prelude functions that read uniforms (shaderflow.glsl:98-146)
    SF_HD vec2 agluv2gluv(const vec2& a) const { return a*vec2(iAspectRatio, 1.0f); }
    SF_HD vec2 gluv2agluv(const vec2& g) const { return g/vec2(iAspectRatio, 1.0f); }
    SF_HD vec2 stuv2stxy(const vec2& s, const vec2& resolution) const { return rt::stuv2stxy(s, resolution); }
    SF_HD vec2 stuv2stxy(const vec2& s) const { return rt::stuv2stxy(s, iResolution); }
    SF_HD vec2 stxy2stuv(const vec2& s, const vec2& resolution) const { return rt::stxy2stuv(s, resolution); }
    SF_HD vec2 stxy2stuv(const vec2& s) const { return rt::stxy2stuv(s, iResolution); }
    SF_HD vec2 astuv2stuv(const vec2& a) const { return vec2(a.x*iAspectRatio + (1.0f - iAspectRatio)/2.0f, a.y); }
    SF_HD vec2 stuv2astuv(const vec2& s) const { return vec2((s.x - (1.0f - iAspectRatio)/2.0f)/iAspectRatio, s.y); }
    SF_HD vec2 gluv_mirrored_repeat(const vec2& g) const { return vec2(iWantAspect*triangle_wave(g.x, 4.0f*iWantAspect), triangle_wave(g.y, 4.0f)); }
    SF_HD bool stuv_oob(const vec2& s) const { return astuv_oob(stuv2astuv(s)); }
    SF_HD bool gluv_oob(const vec2& g) const { return agluv_oob(gluv2agluv(g)); }
    // texture helpers with the mirrored-repeat variants (:172-196)
    SF_HD vec4 gtexture(sampler2D image, const vec2& g) const { return rt::gtexture(image, g); }
    SF_HD vec4 gmtexture(sampler2D image, const vec2& g) const { return rt::gtexture(image, gluv_mirrored_repeat(g)); }
    SF_HD vec4 gtexture(sampler2D image, const vec2& g, bool mirror) const { return mirror ? gmtexture(image, g) : rt::gtexture(image, g); }
    SF_HD vec4 agtexture(sampler2D image, const vec2& a) const { return rt::gtexture(image, agluv2gluv(a)); }
    SF_HD vec4 agmtexture(sampler2D image, const vec2& a) const { return agtexture(image, agluv_mirrored_repeat(a)); }
    SF_HD vec4 agtexture(sampler2D image, const vec2& a, bool mirror) const { return mirror ? agmtexture(image, a) : agtexture(image, a); }

    // GetCamera(iCamera) (camera.glsl:132-155): glsl.hpp's projection, widened to the GLSL struct
    SF_HD Camera get_camera_() const {
        const sf::Camera c = sf::get_camera(*frag_);
        Camera r;
        r.mode = iCameraMode; r.projection = iCameraProjection;
        r.position = up(c.position); r.up = up(c.up); r.down = r.up*(-1.0f); r.right = up(c.right); r.left = r.right*(-1.0f);
        r.forward = up(c.forward); r.backward = up(c.backward); r.zenith = iCameraZenith;
        r.origin = up(c.origin); r.target = up(c.target); r.orbital = c.orbital; r.dolly = c.dolly;
        r.plane_point = vec3(0.0f, 0.0f, 1.0f); r.plane_normal = vec3(0.0f, 0.0f, 1.0f);
        r.gluv = up(c.gluv); r.agluv = up(c.agluv); r.stuv = up(c.stuv); r.astuv = up(c.astuv); r.glxy = up(c.glxy); r.stxy = up(c.stxy);
        r.out_of_bounds = c.out_of_bounds;
        r.separation = c.separation; r.focal_length = c.focal_length; r.isometric = c.isometric; r.zoom = c.zoom;
        return r;
    }
};

}  // namespace rt

// The shader policy of a translated fragment: FRAGMENT is the generated struct (derives from rt::FragmentBase, has
// load_user_() and main_()).
template <class FRAGMENT, bool DERIVATIVES = false, bool TILED = false> struct JitShader : PlainShader<FRAG_DEFAULT> {
    static constexpr bool QUADS = DERIVATIVES;
    SF_HD static vec4 run(const RenderArgs&, const Frag& f, const State&, const Shared&) {
        FRAGMENT s;
        s.load_(f);
        s.load_user_();
        s.main_();
        if (s.discarded_) return {0.0f, 0.0f, 0.0f, 0.0f};        // `discard`: the target keeps its cleared value (no previous content in a fused target)
        return {s.fragColor.x, s.fragColor.y, s.fragColor.z, s.fragColor.w};
    }
};

// The same policy with sampler SF_JIT_TILE_SLOT served from LDS (TileView above). setup(): the threads that own the block's
// first and last sample run the fragment in record mode (the other lanes wait: one extra evaluation per block, against
// FUSED_ROWS x S x S / 4 per lane in the pass proper); the union of the two boxes, cropped about its centre to TILE_TEXELS,
// is decoded into LDS by the whole block. A fragment whose taps move monotonically with the pixel (any blur or feedback
// kernel around the fragment's own position, under any affine map) finds every tap inside; what falls outside is fetched
// the generic way.
#if defined(__HIPCC__)
template <class FRAGMENT> struct JitShader<FRAGMENT, false, true> : PlainShader<FRAG_DEFAULT> {
    static constexpr int ROWS_1X = 4;                // the probe is one evaluation per block: as many per lane as the fused 2x kernel has
    struct State { vec2 agluv; };
    struct Shared {
        __attribute__((aligned(16))) float4 texels[rt::TILE_TEXELS];
        int box[2][4];
    };
    __device__ static void pre(const RenderArgs&, const Frag& f, bool, State& s) { s.agluv = f.agluv; }
    __device__ static int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
    __device__ static int least(int a, int b) { return (b < a) ? b : a; }      // (sf::min / sf::max are the float ones)
    __device__ static int most(int a, int b) { return (a < b) ? b : a; }
    __device__ static void rectangle(const Shared& sh, int& x0, int& y0, int& w, int& h) {
        const int lo_x = uniform(least(sh.box[0][0], sh.box[1][0])), lo_y = uniform(least(sh.box[0][1], sh.box[1][1]));
        const int hi_x = uniform(most(sh.box[0][2], sh.box[1][2])), hi_y = uniform(most(sh.box[0][3], sh.box[1][3]));
        x0 = lo_x; y0 = lo_y; w = hi_x - lo_x + 1; h = hi_y - lo_y + 1;
        if (w <= 0 || h <= 0) { w = 0; h = 0; return; }             // the probes took no tap
        constexpr int WIDEST = rt::TILE_TEXELS/4;
        if (w > WIDEST) { x0 += (w - WIDEST)/2; w = WIDEST; }
        const int fit = rt::TILE_TEXELS/w;                           // (a probe that tapped far away on both sides: h up to 2^25, w*h would overflow)
        if (h > fit) { y0 += (h - fit)/2; h = fit; }
    }
    template <int N> __device__ static void setup(const RenderArgs&, const Tex*, const Frag& f, State (&state)[N], const bool (&)[N], Shared& sh, int corner_tid) {
        const int tid = threadIdx.y*blockDim.x + threadIdx.x, nthreads = blockDim.x*blockDim.y;
        if (tid == 0 || tid == corner_tid) {
            Frag g = f;
            varyings_from_agluv(g, (tid == 0) ? state[0].agluv : state[N - 1].agluv, f.aspect);
            FRAGMENT s;
            s.load_(g);
            int box[4] = {1 << 30, 1 << 30, -(1 << 30), -(1 << 30)};      // a local of its own: registers once `s` is taken apart
            s.tile_.texels = sh.texels; s.tile_.record = box;
            s.load_user_();
            s.main_();
            int* shared_box = sh.box[(tid == 0) ? 0 : 1];
            for (int k = 0; k < 4; k++) shared_box[k] = box[k];
            if (corner_tid == 0) for (int k = 0; k < 4; k++) sh.box[1][k] = box[k];
        }
        __syncthreads();
        int x0, y0, w, h;
        rectangle(sh, x0, y0, w, h);
        const Tex& t = (SF_JIT_TILE_SLOT < TEX_HISTORY) ? f.tex[SF_JIT_TILE_SLOT] : f.history[SF_JIT_TILE_SLOT - TEX_HISTORY];
        if (t.data) {
            for (int k = tid; k < w*h; k += nthreads) {
                const int y = k/w, x = k - y*w;
                const vec4 c = texel(t, wrap_texel(x0 + x, t.width, t.repeat_x), wrap_texel(y0 + y, t.height, t.repeat_y));
                sh.texels[k] = make_float4(c.x, c.y, c.z, c.w);
            }
        }
        __syncthreads();
    }
    __device__ static vec4 run(const RenderArgs&, const Frag& f, const State&, const Shared& sh) {
        FRAGMENT s;
        s.load_(f);
        int x0, y0, w, h;
        rectangle(sh, x0, y0, w, h);
        s.tile_.texels = sh.texels; s.tile_.record = nullptr;
        s.tile_.x0 = x0; s.tile_.y0 = y0; s.tile_.w = w;
        s.tile_.nearest_w = (unsigned)w; s.tile_.nearest_h = (unsigned)h;
        s.tile_.linear_w = (unsigned)most(w - 1, 0); s.tile_.linear_h = (unsigned)most(h - 1, 0);
        s.load_user_();
        s.main_();
        if (s.discarded_) return {0.0f, 0.0f, 0.0f, 0.0f};
        return {s.fragColor.x, s.fragColor.y, s.fragColor.z, s.fragColor.w};
    }
};
#endif

}  // namespace sf

// Host-side execution of a translated fragment, for tests only (tests/test_host_translate.py builds the translation unit as a
// shared library with -DSF_JIT_HOST and compares its pixels with the parity oracle; no product path defines SF_JIT_HOST).
#ifdef SF_JIT_HOST
#include "uniform_table.hpp"
extern "C" {
inline size_t sfx_jit_host_uniforms_size() { return sizeof(sf::Uniforms); }
inline size_t sfx_jit_host_textures_size() { return sizeof(sf::Tex)*sf::TEX_SLOTS; }
inline void sfx_jit_host_defaults(sf::Uniforms* u) { sf::default_uniforms(*u); }
inline int sfx_jit_host_uniform(sf::Uniforms* u, const char* name, const float* values, int count) {
    for (const auto& f : sf::g_uniform_fields) {
        if (strcmp(f.name, name)) continue;
        for (int k = 0; k < count && k < f.count; k++) {
            if (f.integer) ((int*)((char*)u + f.offset))[k] = (int)values[k];
            else ((float*)((char*)u + f.offset))[k] = values[k];
        }
        return 1;
    }
    return 0;
}
inline void sfx_jit_host_user(sf::Uniforms* u, int slot, const void* words, int count) { memcpy(&u->user[slot], words, 4*(size_t)count); }
inline void sfx_jit_host_texture(sf::Tex* textures, int slot, const void* data, int width, int height, int components, int dtype, int filter, int repeat_x, int repeat_y) {
    textures[slot] = sf::Tex{data, width, height, components, dtype, filter, repeat_x, repeat_y};
}
}
#define SF_JIT_HOST_POINTS(FRAGMENT) \
    extern "C" void sfx_jit_host_render(const sf::Uniforms* u, const sf::Tex* textures, int wr, int hr, unsigned* rgba8) { \
        sf::RenderArgs a{}; \
        const float aspect = u->iResolution[0]/u->iResolution[1]; \
        for (int j = 0; j < hr; j++) for (int i = 0; i < wr; i++) { \
            sf::Frag f; f.u = u; f.tex = textures; f.history = textures + sf::TEX_HISTORY; \
            sf::make_varyings(f, i, j, wr, hr, aspect); \
            rgba8[(size_t)j*wr + i] = sf::pack_rgba8(sf::JitShader<FRAGMENT>::run(a, f, {}, {})); \
        } \
    } \
    extern "C" void sfx_jit_host_render_float(const sf::Uniforms* u, const sf::Tex* textures, int wr, int hr, float* rgba) { \
        sf::RenderArgs a{}; \
        const float aspect = u->iResolution[0]/u->iResolution[1]; \
        for (int j = 0; j < hr; j++) for (int i = 0; i < wr; i++) { \
            sf::Frag f; f.u = u; f.tex = textures; f.history = textures + sf::TEX_HISTORY; \
            sf::make_varyings(f, i, j, wr, hr, aspect); \
            const sf::vec4 c = sf::JitShader<FRAGMENT>::run(a, f, {}, {}); \
            float* p = rgba + ((size_t)j*wr + i)*4; p[0] = c.x; p[1] = c.y; p[2] = c.z; p[3] = c.w; \
        } \
    } \
    extern "C" void* sfx_jit_host_keep[] = {(void*)sfx_jit_host_uniforms_size, (void*)sfx_jit_host_textures_size, (void*)sfx_jit_host_defaults, \
                                            (void*)sfx_jit_host_uniform, (void*)sfx_jit_host_user, (void*)sfx_jit_host_texture};
#else
#define SF_JIT_HOST_POINTS(FRAGMENT)
#endif

// Entry points of a code object (capi: sfx_program_load looks them up by these names)
// sfx_jit_flags: bit 0 = SF_JIT_DERIVATIVES, bits 8-15 = rows a lane walks in sfx_jit_render / sfx_jit_fused_1 (the launch geometry).
// SF_JIT_DERIVATIVES (0/1, defined by the translator before this macro): the fragment calls dFdx/dFdy/fwidth — the unfused kernel
// uses the quad layout and the library keeps the program off the fused kernels except at ssaa 2, where the four supersamples of
// a pixel already are the four lanes of a quad (all valid or all invalid together).
// SF_JIT_TILED: the translator asked for an LDS tile (SF_JIT_TILE_SLOT) and the fragment takes no derivatives (a probe with
// two live lanes has no neighbours to difference with).
#define SF_JIT_TILED ((SF_JIT_TILE_SLOT >= 0) && !(SF_JIT_DERIVATIVES))
// sfx_jit_render_quads: the tiled policy walks ROWS_1X rows per lane, so its lanes cannot be laid out as 2 x 2 quads; when a MIPMAPPED
// texture is bound at run time (RenderArgs.quads: the level of detail needs the quad's differences, glsl.hpp texture_mipmapped) the
// library launches this untiled twin instead, whose 64 x 4 blocks take the quad layout (render_kernels.hpp render_body). Untiled
// code objects need none: their sfx_jit_render is that kernel already.
#if SF_JIT_TILE_SLOT >= 0 && !(SF_JIT_DERIVATIVES)
#define SF_JIT_QUADS_ENTRY(FRAGMENT) \
    extern "C" __global__ __launch_bounds__(256) void sfx_jit_render_quads(const sf::RenderArgs a) { sf::render_body<sf::JitShader<FRAGMENT, false, false>>(a); }
#else
#define SF_JIT_QUADS_ENTRY(FRAGMENT)
#endif
#define SF_JIT_ENTRY_POINTS(FRAGMENT) \
    SF_JIT_HOST_POINTS(FRAGMENT) \
    extern "C" __device__ __attribute__((used)) const unsigned long long sfx_jit_layout = sf::render_args_layout(); \
    extern "C" __device__ __attribute__((used)) const unsigned sfx_jit_flags = (SF_JIT_DERIVATIVES ? 1u : 0u) | \
        ((unsigned)sf::shader_rows_1x<sf::JitShader<FRAGMENT, false, SF_JIT_TILED>>::value << 8); \
    extern "C" __global__ __launch_bounds__(256) void sfx_jit_render(const sf::RenderArgs a) { sf::render_body<sf::JitShader<FRAGMENT, (SF_JIT_DERIVATIVES != 0), SF_JIT_TILED>>(a); } \
    SF_JIT_QUADS_ENTRY(FRAGMENT) \
    extern "C" __global__ __launch_bounds__(256) void sfx_jit_fused_1(const sf::RenderArgs a) { sf::render_resolve_body<sf::JitShader<FRAGMENT, false, SF_JIT_TILED>, 1>(a); } \
    extern "C" __global__ __launch_bounds__(512) void sfx_jit_fused_2(const sf::RenderArgs a) { sf::render_resolve_body<sf::JitShader<FRAGMENT, (SF_JIT_DERIVATIVES != 0), SF_JIT_TILED>, 2>(a); } \
    extern "C" __global__ __launch_bounds__(512) void sfx_jit_fused_4(const sf::RenderArgs a) { sf::render_resolve_body<sf::JitShader<FRAGMENT, false, SF_JIT_TILED>, 4>(a); }
