// glsl.hpp — GLSL-semantics device header: vector types, the OpenGL 3.3 sampler, the varyings of the
// reference's vertex stage and the subset of its prelude the in-scope fragments use.
//
// Follows (file:line in the reference): resources/shaders/vertex/default.glsl:1-17,
// resources/shaders/include/shaderflow.glsl (coords :91-99, textures :165-204, zoom :361-367,
// atan :378-392, hsv2rgb :406-425, rotate :75-83), resources/shaders/include/camera.glsl:53-155,
// texture.py:274-283 (sampler state) and OpenGL 3.3 core §3.8.8-3.8.9 / §2.1.6.
//
// Evaluation rules (shared with the parity oracle, written independently there): binary32, left to
// right, no contraction; bilinear weights applied as fma(w11,t11, fma(w01,t01, fma(w10,t10, w00*t00)));
// unorm8 texel = c/255.0f; colour write = clamp (NaN→0), *255, round half to even.
#pragma once

#include "sfmath.hpp"

namespace sf {

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };

SF_HD vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
SF_HD vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
SF_HD vec2 operator*(vec2 a, vec2 b) { return {a.x*b.x, a.y*b.y}; }
SF_HD vec2 operator/(vec2 a, vec2 b) { return {a.x/b.x, a.y/b.y}; }
SF_HD vec2 operator+(vec2 a, float s) { return {a.x + s, a.y + s}; }
SF_HD vec2 operator-(vec2 a, float s) { return {a.x - s, a.y - s}; }
SF_HD vec2 operator*(vec2 a, float s) { return {a.x*s, a.y*s}; }
SF_HD vec2 operator*(float s, vec2 a) { return {s*a.x, s*a.y}; }
SF_HD vec2 operator/(vec2 a, float s) { return {a.x/s, a.y/s}; }
SF_HD vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
SF_HD vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
SF_HD vec3 operator*(vec3 a, float s) { return {a.x*s, a.y*s, a.z*s}; }
SF_HD vec3 operator*(float s, vec3 a) { return {s*a.x, s*a.y, s*a.z}; }
SF_HD vec3 operator+(vec3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
SF_HD vec3 operator/(vec3 a, float s) { return {a.x/s, a.y/s, a.z/s}; }
SF_HD vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
SF_HD vec4 operator*(vec4 a, float s) { return {a.x*s, a.y*s, a.z*s, a.w*s}; }
SF_HD vec4 operator/(vec4 a, float s) { return {a.x/s, a.y/s, a.z/s, a.w/s}; }
SF_HD float dot(vec3 a, vec3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }
SF_HD float length(vec2 a) { return sf::sqrt(a.x*a.x + a.y*a.y); }
SF_HD vec3 mix(vec3 a, vec3 b, float t) { return {sf::mix(a.x, b.x, t), sf::mix(a.y, b.y, t), sf::mix(a.z, b.z, t)}; }
SF_HD vec3 cross(vec3 a, vec3 b) { return {a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y}; }
SF_HD vec3 rgb(vec4 c) { return {c.x, c.y, c.z}; }
SF_HD void set_rgb(vec4& c, vec3 v) { c.x = v.x; c.y = v.y; c.z = v.z; }

// ---- textures ----------------------------------------------------------------------------------

enum : int { DT_U8 = 0, DT_F32 = 1, DT_U16 = 2, DT_F16 = 3 };
// FILTER_LINEAR_MIPMAP / FILTER_NEAREST_MIPMAP: texture.py:131-137 — moderngl.LINEAR_MIPMAP_LINEAR / NEAREST_MIPMAP_NEAREST as the
// minification filter once `mipmaps=True` has built the chain (texture.py:277-278); the magnification filter stays LINEAR / NEAREST
// FILTER_LINEAR_FIXED8: the context's opt-in filter model (sfx_ctx_filter_model, capi.hip tex_view) for LINEAR unorm8 textures — the
// fixed-point bilinear filter of the software rasteriser the reference's CPU path runs on (below: texture_fixed8)
enum : int { FILTER_NEAREST = 0, FILTER_LINEAR = 1, FILTER_LINEAR_MIPMAP = 2, FILTER_NEAREST_MIPMAP = 3, FILTER_LINEAR_FIXED8 = 4 };

struct Tex {                       // one TextureBox of texture.py:56-70, rows bottom-up, tightly packed
    const void* data;
    int width, height, components, dtype, filter, repeat_x, repeat_y;
    const void* mips;              // levels 1 … levels-1, tightly packed one after the other (level k: max(1, width >> k) x max(1, height >> k))
    int levels;                    // 0 or 1: no chain (the mipmap filters then read level 0 only)
};

SF_HD size_t texel_bytes(const Tex& t) { return (size_t)t.components*(t.dtype == DT_U8 ? 1 : (t.dtype == DT_F32 ? 4 : 2)); }
// level k of a chain as a texture of its own (k = 0: the texture itself)
SF_HD Tex mip_level(const Tex& t, int k) {
    Tex v = t;
    if (k <= 0 || t.levels <= 1 || !t.mips) return v;
    if (k > t.levels - 1) k = t.levels - 1;
    size_t offset = 0;
    int w = t.width, h = t.height;
    for (int l = 1; l <= k; l++) {
        w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1;
        if (l < k) offset += (((size_t)w*h*texel_bytes(t)) + 15) & ~(size_t)15;     // every level starts 16-byte aligned (+ the RGB8 over-read)
    }
    v.data = (const char*)t.mips + offset; v.width = w; v.height = h;
    return v;
}

SF_HD int wrap_texel(int i, int size, int repeat) {
    if ((unsigned)i < (unsigned)size) return i;    // inside the texture: no integer division
    if (repeat) { int m = i % size; return (m < 0) ? m + size : m; }
    return (i < 0) ? 0 : ((i >= size) ? size - 1 : i);
}

// c/255.0f for an integer-valued 0 <= c <= 255 (a texel byte) without the IEEE division sequence: 1/255 split into its float
// and the float of the remainder, c*hi + c*lo with ONE rounding of the sum — the correctly rounded quotient for all 256 bytes
// (tools/check_unorm8.py does the exact rational arithmetic), so the result is bit-identical to `c/255.0f` (the parity oracle
// writes the division). Two operations per component; every RGBA8 tap of the generic sampler converts sixteen of them.
SF_HD float unorm8_to_float(float c) {
    return fmaf(c, 0x1.010102p-8f, c*-0x1.fdfdfep-33f);
}

SF_HD vec4 texel(const Tex& t, int i, int j) {
    // written without a local array: a dynamically indexed float[4] would be promoted to LDS
    vec4 c = {0.0f, 0.0f, 0.0f, 1.0f};
    const uint32_t index = (uint32_t)j*(uint32_t)t.width + (uint32_t)i;      // texel index: 0 <= i < width, 0 <= j < height (callers wrap first); < 2^32 texels
    const int n = t.components;
    const size_t base = (size_t)index*(size_t)n;
    if (t.dtype == DT_U8 && n == 4) {                              // RGBA8 (iScreen, history textures): one aligned 32-bit load
        const uint32_t w = ((const uint32_t*)t.data)[index];
        c.x = unorm8_to_float((float)(w & 255u));
        c.y = unorm8_to_float((float)((w >> 8) & 255u));
        c.z = unorm8_to_float((float)((w >> 16) & 255u));
        c.w = unorm8_to_float((float)(w >> 24));
#if defined(__HIP_DEVICE_COMPILE__)
    } else if (t.dtype == DT_U8 && n == 3) {                          // RGB8 (backgrounds, video frames): one unaligned 32-bit load, the fourth byte ignored
        typedef uint32_t unaligned_u32 __attribute__((aligned(1)));  // (device allocations are padded: the last texel's extra byte exists)
        const uint32_t w = *(const unaligned_u32*)((const uint8_t*)t.data + base);
        c.x = unorm8_to_float((float)(w & 255u));
        c.y = unorm8_to_float((float)((w >> 8) & 255u));
        c.z = unorm8_to_float((float)((w >> 16) & 255u));
#endif
    } else if (t.dtype == DT_U8) {
        const uint8_t* p = (const uint8_t*)t.data + base;
        c.x = unorm8_to_float((float)p[0]);
        if (n > 1) c.y = unorm8_to_float((float)p[1]);
        if (n > 2) c.z = unorm8_to_float((float)p[2]);
        if (n > 3) c.w = unorm8_to_float((float)p[3]);
    } else if (t.dtype == DT_F32) {
        const float* p = (const float*)t.data + base;
        c.x = p[0];
        if (n > 1) c.y = p[1];
        if (n > 2) c.z = p[2];
        if (n > 3) c.w = p[3];
    } else if (t.dtype == DT_U16) {
        const uint16_t* p = (const uint16_t*)t.data + base;
        c.x = (float)p[0]/65535.0f;
        if (n > 1) c.y = (float)p[1]/65535.0f;
        if (n > 2) c.z = (float)p[2]/65535.0f;
        if (n > 3) c.w = (float)p[3]/65535.0f;
    } else if (t.dtype == DT_F16) {                               // numpy float16 = IEEE binary16 ("f2", texture.py:28-38)
        const _Float16* p = (const _Float16*)t.data + base;
        c.x = (float)p[0];
        if (n > 1) c.y = (float)p[1];
        if (n > 2) c.z = (float)p[2];
        if (n > 3) c.w = (float)p[3];
    }
    return c;
}

SF_HD float bilerp(float w00, float w10, float w01, float w11, float t00, float t10, float t01, float t11) {
    return fmaf(w11, t11, fmaf(w01, t01, fmaf(w10, t10, w00*t00)));
}

// (out of line: the mipmapped path is rare, and inlined into every tap of every fragment it cost the plain kernels their registers —
// k_render<dynamics> at 1080p: 0.19 → 1.17 ms)
#define SF_COLD static __host__ __device__ __attribute__((noinline))
SF_COLD vec4 texture_mipmapped(Tex t, vec2 uv);      // by value: a reference would pin every caller's Tex in scratch memory
SF_HD vec4 texture(const Tex& t, vec2 uv) {       // GLSL texture(sampler2D, vec2)
    if (__builtin_expect(t.filter >= FILTER_LINEAR_MIPMAP, 0)) return texture_mipmapped(t, uv);
    float u = uv.x*(float)t.width;
    float v = uv.y*(float)t.height;
    if (t.filter == FILTER_NEAREST) {
        int i = wrap_texel((int)::floorf(u), t.width, t.repeat_x);
        int j = wrap_texel((int)::floorf(v), t.height, t.repeat_y);
        return texel(t, i, j);
    }
    float ub = u - 0.5f, vb = v - 0.5f;
    float fu = ::floorf(ub), fv = ::floorf(vb);
    float a = ub - fu, b = vb - fv;
    int i0 = wrap_texel((int)fu, t.width, t.repeat_x), i1 = wrap_texel((int)fu + 1, t.width, t.repeat_x);
    int j0 = wrap_texel((int)fv, t.height, t.repeat_y), j1 = wrap_texel((int)fv + 1, t.height, t.repeat_y);
    vec4 t00 = texel(t, i0, j0), t10 = texel(t, i1, j0), t01 = texel(t, i0, j1), t11 = texel(t, i1, j1);
    float na = 1.0f - a, nb = 1.0f - b;
    float w00 = na*nb, w10 = a*nb, w01 = na*b, w11 = a*b;
    return {bilerp(w00, w10, w01, w11, t00.x, t10.x, t01.x, t11.x),
            bilerp(w00, w10, w01, w11, t00.y, t10.y, t01.y, t11.y),
            bilerp(w00, w10, w01, w11, t00.z, t10.z, t01.z, t11.z),
            bilerp(w00, w10, w01, w11, t00.w, t10.w, t01.w, t11.w)};
}

// ---- mipmapped sampling (OpenGL 3.3 core section 3.8.11; texture.py:131-137, 277-278) ----------------------------------------------
// Level of detail from the implicit derivatives of the coordinate: the kernels that may meet such a texture lay their lanes out as
// 2 x 2 quads (render_kernels.hpp: lane bit 0 = x, bit 1 = y inside the quad), and the derivative is the difference across the
// quad's row / column — the "fine" form, as dFdx/dFdy of translated fragments. rho is the longer of the two footprint axes in texels
// of level 0 (equation 3.21's ideal scale factor), lambda = log2(rho), no bias (the reference sets none), clamped to the chain.
// lambda <= 0 magnifies: the plain LINEAR / NEAREST filter on level 0. LINEAR_MIPMAP_LINEAR blends the bilinear samples of levels
// floor(lambda) and floor(lambda)+1 by frac(lambda); NEAREST_MIPMAP_NEAREST takes the nearest texel of level ceil(lambda + 0.5) - 1.
// What an implementation may approximate here it does (llvmpipe: log2 by mantissa, 8-bit blend — the oracle's llvmpipe switch has its
// arithmetic, tests/golden/mip.npz its frames); these are the specification's own formulas in binary32.
SF_HD float quad_dx(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0xF5, 0xf, 0xf, true)) - __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0xA0, 0xf, 0xf, true));
#else
    (void)v; return 0.0f;                                            // host builds (tests of translated fragments) read level 0
#endif
}
SF_HD float quad_dy(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0xEE, 0xf, 0xf, true)) - __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0x44, 0xf, 0xf, true));
#else
    (void)v; return 0.0f;
#endif
}
SF_HD vec4 texture_level(const Tex& t, vec2 uv, int level, int filter) {
    Tex v = mip_level(t, level);
    v.filter = filter;
    return texture(v, uv);
}
SF_HD float mip_lambda(const Tex& t, float dudx, float dvdx, float dudy, float dvdy) {
    const float along_x = dudx*dudx + dvdx*dvdx, along_y = dudy*dudy + dvdy*dvdy;
    const float rho2 = along_x > along_y ? along_x : along_y;
    const float lambda = 0.5f*sf::log2(rho2);                        // log2(rho); rho2 = 0 (a constant coordinate) gives -inf: magnification
    const float top = (float)((t.levels > 1 ? t.levels : 1) - 1);
    return lambda > top ? top : lambda;
}
// ---- the fixed-point filter model (opt-in, sfx_ctx_filter_model) ---------------------------------------------------------------------
// OpenGL leaves the precision of the bilinear weights to the implementation (>= 4 subtexel bits, 3.3 core section 3.8.11). The float
// weights above are one admissible choice; Mesa llvmpipe — the software rasteriser `north_star` names as the reference's CPU path, and
// the renderer of tests/golden/mesa*.npz — makes another for unorm8 textures: texel coordinates in 24.8 fixed point, rint(s*N*256) - 128,
// the low 8 bits the weight; a lerp along x on the bytes, a + ((w*(b - a) + 128) >> 8), ROUNDED BACK TO 8 BITS, then the same along y:
// a filtered texel is always k/255. Measured bit for bit on that implementation (tests/golden/filter.npz, tests/test_gpu_filter_model.py;
// the parity oracle keeps the checker's copy). With the model on, frames meet that implementation's within 1 LSB where
// the float weights leave up to 1.3 % of the values 2 off (a second filter or a quantisation after the first: tests/test_gpu_mesa.py).
SF_HD int fixed8_lerp(int w, int a, int b) { return (a + ((w*(b - a) + 128) >> 8)) & 255; }
SF_COLD vec4 texture_fixed8(Tex t, vec2 uv) {
    const int fx = (int)::rintf(uv.x*(float)t.width*256.0f) - 128, fy = (int)::rintf(uv.y*(float)t.height*256.0f) - 128;
    const int wx = fx & 255, wy = fy & 255;
    const int i0 = wrap_texel(fx >> 8, t.width, t.repeat_x), i1 = wrap_texel((fx >> 8) + 1, t.width, t.repeat_x);
    const int j0 = wrap_texel(fy >> 8, t.height, t.repeat_y), j1 = wrap_texel((fy >> 8) + 1, t.height, t.repeat_y);
    const uint8_t* data = (const uint8_t*)t.data;
    const int n = t.components;
    const uint8_t* p00 = data + ((size_t)j0*t.width + i0)*n; const uint8_t* p10 = data + ((size_t)j0*t.width + i1)*n;
    const uint8_t* p01 = data + ((size_t)j1*t.width + i0)*n; const uint8_t* p11 = data + ((size_t)j1*t.width + i1)*n;
    auto channel = [&](int k) { return unorm8_to_float((float)fixed8_lerp(wy, fixed8_lerp(wx, p00[k], p10[k]), fixed8_lerp(wx, p01[k], p11[k]))); };
    vec4 c = {channel(0), 0.0f, 0.0f, 1.0f};
    if (n > 1) c.y = channel(1);
    if (n > 2) c.z = channel(2);
    if (n > 3) c.w = channel(3);
    return c;
}

SF_COLD vec4 texture_mipmapped(Tex t, vec2 uv) {
    if (t.filter == FILTER_LINEAR_FIXED8) return texture_fixed8(t, uv);            // (shares the cold entry: one rare branch in texture())
    const float u = uv.x*(float)t.width, v = uv.y*(float)t.height;
    const float lambda = mip_lambda(t, quad_dx(u), quad_dx(v), quad_dy(u), quad_dy(v));
    const bool linear = (t.filter == FILTER_LINEAR_MIPMAP);
    if (!(lambda > 0.0f) || t.levels <= 1) return texture_level(t, uv, 0, linear ? FILTER_LINEAR : FILTER_NEAREST);
    if (!linear) return texture_level(t, uv, (int)::ceilf(lambda + 0.5f) - 1, FILTER_NEAREST);
    const float below = ::floorf(lambda), f = lambda - below;
    const int d1 = (int)below, d2 = d1 + 1 < t.levels ? d1 + 1 : t.levels - 1;
    const vec4 a = texture_level(t, uv, d1, FILTER_LINEAR);
    if (d2 == d1 || f == 0.0f) return a;
    const vec4 b = texture_level(t, uv, d2, FILTER_LINEAR);
    return {fmaf(f, b.x - a.x, a.x), fmaf(f, b.y - a.y, a.y), fmaf(f, b.z - a.z, a.z), fmaf(f, b.w - a.w, a.w)};
}

// texture(t, uv).xy for fragments that read two components only: the same operations as texture() on .x and .y
// (so the same bits), without blending the two components the caller drops
SF_HD vec2 texel_xy(const Tex& t, int i, int j) {
    vec4 c = texel(t, i, j);
    return {c.x, c.y};
}
SF_HD vec2 texture_xy(const Tex& t, vec2 uv) {
    if (t.filter >= FILTER_LINEAR_MIPMAP) { const vec4 c = texture_mipmapped(t, uv); return {c.x, c.y}; }
    float u = uv.x*(float)t.width;
    float v = uv.y*(float)t.height;
    if (t.filter == FILTER_NEAREST) {
        int i = wrap_texel((int)::floorf(u), t.width, t.repeat_x);
        int j = wrap_texel((int)::floorf(v), t.height, t.repeat_y);
        return texel_xy(t, i, j);
    }
    float ub = u - 0.5f, vb = v - 0.5f;
    float fu = ::floorf(ub), fv = ::floorf(vb);
    float a = ub - fu, b = vb - fv;
    int i0 = wrap_texel((int)fu, t.width, t.repeat_x), i1 = wrap_texel((int)fu + 1, t.width, t.repeat_x);
    int j0 = wrap_texel((int)fv, t.height, t.repeat_y), j1 = wrap_texel((int)fv + 1, t.height, t.repeat_y);
    vec2 t00 = texel_xy(t, i0, j0), t10 = texel_xy(t, i1, j0), t01 = t00, t11 = t10;
    if (j1 != j0) { t01 = texel_xy(t, i0, j1); t11 = texel_xy(t, i1, j1); }      // one-row textures (iWaveform): both rows are row 0
    float na = 1.0f - a, nb = 1.0f - b;
    float w00 = na*nb, w10 = a*nb, w01 = na*b, w11 = a*b;
    return {bilerp(w00, w10, w01, w11, t00.x, t10.x, t01.x, t11.x), bilerp(w00, w10, w01, w11, t00.y, t10.y, t01.y, t11.y)};
}

// texelFetch(sampler, ivec2, 0). Out-of-range coordinates are undefined in GL 3.3 (§3.8.9); with robust buffer
// access — what desktop drivers do in practice — the fetch returns zeros, and that is the rule here.
SF_HD vec4 texel_fetch(const Tex& t, int i, int j) {
    if ((unsigned)i >= (unsigned)t.width || (unsigned)j >= (unsigned)t.height) return {0.0f, 0.0f, 0.0f, 0.0f};
    return texel(t, i, j);
}

// Colour write to a unorm8 target (OpenGL 3.3 §2.1.6 / §4.1): clamp to [0, 1] (NaN -> 0), scale by 255, round half to even.
// On the device that is ONE instruction after the multiplication: v_cvt_pk_u8_f32 clamps to [0, 255], rounds to nearest even,
// maps NaN to 0 and writes the byte into its place of the packed texel — identical to the sequence below for every one of the
// 2^32 floats (tools/check_cvt_pk_u8.hip, run on gfx950), and clamp(c)*255 == clamp(c*255) bit for bit.
SF_HD uint32_t unorm8(float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_cvt_pk_u8_f32(c*255.0f, 0u, 0u);
#else
    c = (c > 0.0f) ? c : 0.0f;
    c = (c < 1.0f) ? c : 1.0f;
    return (uint32_t)::rintf(c*255.0f);
#endif
}
SF_HD uint32_t pack_rgba8(vec4 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t texel = __builtin_amdgcn_cvt_pk_u8_f32(c.x*255.0f, 0u, 0u);
    texel = __builtin_amdgcn_cvt_pk_u8_f32(c.y*255.0f, 1u, texel);
    texel = __builtin_amdgcn_cvt_pk_u8_f32(c.z*255.0f, 2u, texel);
    return __builtin_amdgcn_cvt_pk_u8_f32(c.w*255.0f, 3u, texel);
#else
    return unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24);
#endif
}
SF_HD uint32_t pack_rgb8(vec3 c) {                                  // the fused kernels never look at alpha (final.glsl:6-31 takes .rgb)
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t texel = __builtin_amdgcn_cvt_pk_u8_f32(c.x*255.0f, 0u, 0u);
    texel = __builtin_amdgcn_cvt_pk_u8_f32(c.y*255.0f, 1u, texel);
    return __builtin_amdgcn_cvt_pk_u8_f32(c.z*255.0f, 2u, texel);
#else
    return unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16);
#endif
}

// ---- uniforms ----------------------------------------------------------------------------------
// scene.py:687-703, camera.py:196-201 (+ its ShaderDynamics :147-185), audio/module.py:413-421,
// spectrogram.py:313-320, waveform.py:89-90. Host fills it by name (capi: sfx_uniform_set).

constexpr int USER_SLOTS = 64;
struct Uniforms {
    float iTime, iTau, iDuration, iDeltatime;
    float iResolution[2];
    float iWantAspect, iQuality, iSSAA, iFramerate;
    int iFrame, iRealtime, iLayer, iSubsample;
    float iMouse[2];
    int iMouseInside, iMouse1, iMouse2;
    int iCameraMode, iCameraProjection;
    float iCameraRight[3], iCameraUpward[3], iCameraForward[3];
    float iCameraPosition[3], iCameraZenith[3];
    float iCameraSeparation, iCameraZoom, iCameraIsometric, iCameraFocalLength, iCameraOrbital, iCameraDolly;
    float iAudioVolume, iAudioVolumeIntegral, iAudioSTD;
    int iSpectrogramLength, iSpectrogramBins, iSpectrogramSmooth, iSpectrogramScroll;
    float iSpectrogramOffset, iSpectrogramMin, iSpectrogramMax;
    int iWaveformLength;
    float user[USER_SLOTS];        // scene-defined uniforms: fixed slots for the restated fragments (fragments.hpp), in declaration order for translated ones
};

// Sampler slots: four named textures, then the temporal history of the texture a fragment reads by
// `<name>{t}x0` (texture.py:346-347, 380-381): slot TEX_HISTORY + t is `t` frames back, layer 0.
enum : int { TEX_BACKGROUND = 0, TEX_SPECTROGRAM = 1, TEX_WAVEFORM = 2, TEX_CHILD = 3, TEX_HISTORY = 4, TEX_HISTORY_DEPTH = 12, TEX_SLOTS = 16 };

// Per-frame values that live on the device in tape (batched export) mode; written by the dynamics scan
struct FrameDyn {
    float iTime, iTau, iAudioVolume, iAudioVolumeIntegral, iAudioSTD, iSpectrogramOffset;
    int iFrame, pad;
};

// ---- varyings (vertex/default.glsl:1-17) ---------------------------------------------------------

struct Frag {
    const Uniforms* u;
    const Tex* tex;                                // the named slots [0, TEX_HISTORY)
    const Tex* history;                            // history[t] = slot TEX_HISTORY + t (read in place from the kernel arguments)
    vec2 agluv, gluv, astuv, stuv, stxy, glxy, fragCoord;
    float aspect;                                  // iAspectRatio, shaderflow.glsl:16
};

SF_HD vec2 gluv2stuv(vec2 g) { return (g + 1.0f)/2.0f; }          // shaderflow.glsl:95
SF_HD vec2 stuv2gluv(vec2 s) { return (s*2.0f) - 1.0f; }          // shaderflow.glsl:91

// (i + 0.5)/n, the pixel centre of vertex/default.glsl's interpolation. `inv` != 0 is RN(1/n) from a host that has checked, for
// EVERY i in [0, n), that the Newton-corrected product below equals the IEEE quotient bit for bit (capi: pixel_centre_reciprocal):
// three plain operations instead of the division sequence, same bits. `inv` == 0: the division.
SF_HD float pixel_centre(int i, int n, float inv) {
    const float c = (float)i + 0.5f;
    if (inv != 0.0f) { const float q = c*inv; return fmaf(fmaf(-q, (float)n, c), inv, q); }
    return c/(float)n;
}

// `aspect` = iResolution.x/iResolution.y, the same IEEE division done once by the caller
// (every varying is a function of agluv: a kernel that has to come back to a sample keeps that one vec2)
SF_HD void varyings_from_agluv(Frag& f, vec2 agluv, float aspect) {
    const Uniforms& u = *f.u;
    vec2 res = {u.iResolution[0], u.iResolution[1]};
    f.aspect = aspect;
    f.agluv = agluv;
    f.gluv = f.agluv*vec2{f.aspect, 1.0f};                         // agluv2gluv, shaderflow.glsl:99
    f.astuv = gluv2stuv(f.agluv);
    f.stuv = gluv2stuv(f.gluv);
    f.stxy = (res*f.astuv) + 1.0f;
    f.glxy = f.stxy - res/2.0f;
    f.fragCoord = f.stxy;
}
SF_HD void make_varyings(Frag& f, int i, int j, int wr, int hr, float aspect, float inv_wr = 0.0f, float inv_hr = 0.0f) {
    vec2 centre = {pixel_centre(i, wr, inv_wr), pixel_centre(j, hr, inv_hr)};
    varyings_from_agluv(f, centre*2.0f - 1.0f, aspect);
}

// ---- prelude subset ------------------------------------------------------------------------------

SF_HD vec2 rotate2d_apply(float angle, vec2 p) {                   // mat2(c,-s,s,c)*p, shaderflow.glsl:75-77
    float c = sf::cos(angle), s = sf::sin(angle);
    return {c*p.x + s*p.y, (-s)*p.x + c*p.y};
}
SF_HD vec2 zoom(vec2 uv, float z, vec2 anchor) { return (uv - anchor)*(z*z) + anchor; }   // :361-363
SF_HD vec4 gtexture(const Tex& t, vec2 gluv) {                     // :165-169
    vec2 scale = {(float)t.height/(float)t.width, 1.0f};
    return texture(t, gluv2stuv(gluv*scale));
}
SF_HD vec4 stexture(const Tex& t, vec2 stuv) { return gtexture(t, stuv2gluv(stuv)); }     // :198-200
SF_HD float atan1n(vec2 p) { return sf::atan(p.y, p.x)/PI; }                               // :378-380
SF_HD float atan2(float y, float x) {                                                     // :382-388
    if (y < 0.0f) return TAU - sf::atan(-y, x);
    return sf::atan(y, x);
}
SF_HD vec3 hsv2rgb(float h, float s, float v) {                                            // :406-425
    h = sf::mod(h, TAU);
    float c = v*s;
    float x = c*(1.0f - sf::abs(sf::mod(h/(PI/3.0f), 2.0f) - 1.0f));
    float m = v - c;
    vec3 rgb;
    switch (sf::to_int(::floorf(6.0f*(h/(2.0f*PI))))) {
        case 0: rgb = {c, x, 0.0f}; break;
        case 1: rgb = {x, c, 0.0f}; break;
        case 2: rgb = {0.0f, c, x}; break;
        case 3: rgb = {0.0f, x, c}; break;
        case 4: rgb = {x, 0.0f, c}; break;
        case 5: rgb = {c, 0.0f, x}; break;
        default: rgb = {0.0f, 0.0f, 0.0f};
    }
    return rgb + m;
}
SF_HD vec3 palette(float t, vec3 A, vec3 B, vec3 C, vec3 D) {                                // :210-218
    if (t < 0.25f) return mix(A, B, t*4.0f);
    if (t < 0.5f) return mix(B, C, (t - 0.25f)*4.0f);
    return mix(C, D, (t - 0.5f)*4.0f);
}
SF_HD vec3 rotate3d(vec3 v, vec3 axis, float angle) {                                      // :81-83
    float c = sf::cos(angle), s = sf::sin(angle);
    return mix(dot(axis, v)*axis, v, c) + cross(axis, v)*s;
}

// ---- camera (camera.glsl) ------------------------------------------------------------------------

struct Camera {
    vec3 position, up, right, forward, backward, origin, target;
    float orbital, dolly, separation, focal_length, isometric, zoom;
    vec2 gluv, agluv, stuv, astuv, glxy, stxy;
    bool out_of_bounds;
};

SF_HD vec3 camera_rectangle(const Camera& c, vec2 g, float size) { return size*(g.x*c.right + g.y*c.up); }   // :54-56
SF_HD vec3 camera_ray_origin(const Camera& c, vec2 g) {                                                      // :58-63
    return c.position + camera_rectangle(c, g, c.zoom*c.isometric) + (c.backward*c.orbital) + (c.backward*c.dolly);
}
SF_HD vec3 camera_ray_target(const Camera& c, vec2 g) {                                                      // :65-70
    return c.position + camera_rectangle(c, g, c.zoom) + (c.backward*c.orbital) + (c.forward*c.focal_length);
}

// The freshly built camera (camera.py:147-185: position 0, identity basis, zoom 1, isometric 0, focal length 1,
// orbital 0, dolly 0, perspective) projects every fragment onto itself: origin = 0, target = (gluv, 1), t = 1, and
// every product with 0 or 1 is exact, so iCamera.gluv == gluv bit for bit. Hosts may set `identity_camera`.
SF_HD bool camera_is_identity(const Uniforms& u) {
    return u.iCameraProjection == 0
        && u.iCameraPosition[0] == 0.0f && u.iCameraPosition[1] == 0.0f && u.iCameraPosition[2] == 0.0f
        && u.iCameraRight[0] == 1.0f && u.iCameraRight[1] == 0.0f && u.iCameraRight[2] == 0.0f
        && u.iCameraUpward[0] == 0.0f && u.iCameraUpward[1] == 1.0f && u.iCameraUpward[2] == 0.0f
        && u.iCameraForward[0] == 0.0f && u.iCameraForward[1] == 0.0f && u.iCameraForward[2] == 1.0f
        && u.iCameraZoom == 1.0f && u.iCameraIsometric == 0.0f && u.iCameraFocalLength == 1.0f
        && u.iCameraOrbital == 0.0f && u.iCameraDolly == 0.0f;
}

// A perspective camera whose basis is the untouched one (right = x, up = y, forward = z — any position, zoom, isometric factor,
// focal length, orbital and dolly distance): every product of the ray construction that mixes the two screen coordinates is a
// product with an exact 0, so iCamera.gluv.x is a function of gluv.x alone and iCamera.gluv.y of gluv.y alone (up to the sign of a
// zero), and `t` of CameraRay2D is the same for every fragment. camera_along_axis() evaluates get_camera for one coordinate.
SF_HD bool camera_is_axis_aligned(const Uniforms& u) {
    return u.iCameraProjection == 0                                // (the stereoscopic projection is separable too, but jumps at the centre column: a block's window is no longer bounded by its corner samples)
        && u.iCameraRight[0] == 1.0f && u.iCameraRight[1] == 0.0f && u.iCameraRight[2] == 0.0f
        && u.iCameraUpward[0] == 0.0f && u.iCameraUpward[1] == 1.0f && u.iCameraUpward[2] == 0.0f
        && u.iCameraForward[0] == 0.0f && u.iCameraForward[1] == 0.0f && u.iCameraForward[2] == 1.0f;
}
SF_HD Camera get_camera(const Frag& f) {                           // GetCamera :132-155 → CameraProject :93-130
    const Uniforms& u = *f.u;
    Camera c;
    const vec3 plane_point = {0.0f, 0.0f, 1.0f}, plane_normal = {0.0f, 0.0f, 1.0f};
    c.position = {u.iCameraPosition[0], u.iCameraPosition[1], u.iCameraPosition[2]};
    c.orbital = u.iCameraOrbital;
    c.dolly = u.iCameraDolly;
    c.up = {u.iCameraUpward[0], u.iCameraUpward[1], u.iCameraUpward[2]};
    c.right = {u.iCameraRight[0], u.iCameraRight[1], u.iCameraRight[2]};
    c.forward = {u.iCameraForward[0], u.iCameraForward[1], u.iCameraForward[2]};
    c.backward = c.forward*(-1.0f);
    c.isometric = u.iCameraIsometric;
    c.focal_length = u.iCameraFocalLength;
    c.zoom = u.iCameraZoom;
    c.separation = u.iCameraSeparation;

    if (u.iCameraProjection == 0) {                                // perspective
        c.origin = camera_ray_origin(c, f.gluv);
        c.target = camera_ray_target(c, f.gluv);
    } else if (u.iCameraProjection == 1) {                         // stereoscopic
        float side = sf::sign(f.agluv.x);
        vec2 g = f.gluv - side*vec2{f.aspect/2.0f, 0.0f};
        c.position = c.position + (side*c.separation)*c.right;
        c.origin = camera_ray_origin(c, g);
        c.target = camera_ray_target(c, g);
    } else {                                                       // equirectangular
        float inclination = c.zoom*(PI*f.agluv.y/2.0f);
        float azimuth = c.zoom*(PI*f.agluv.x/1.0f);
        vec3 target = c.forward;
        target = rotate3d(target, c.right, -inclination);
        target = rotate3d(target, c.up, azimuth);
        c.origin = c.position;
        c.target = c.position + target;
    }

    // CameraRay2D :73-91
    float num = dot(plane_point - c.origin, plane_normal);
    float den = dot(c.target - c.origin, plane_normal);
    float t = num/den;
    c.out_of_bounds = (t < 0.0f) || (sf::abs(f.gluv.x) > u.iWantAspect);
    vec3 hit = c.origin + ((c.target - c.origin)*t);
    vec2 res = {u.iResolution[0], u.iResolution[1]};
    c.gluv = {hit.x, hit.y};
    c.agluv = c.gluv/vec2{f.aspect, 1.0f};
    c.stuv = (c.gluv + 1.0f)/2.0f;
    c.astuv = (c.agluv + 1.0f)/2.0f;
    c.stxy = res*c.astuv;
    c.glxy = c.stxy - res/2.0f;
    return c;
}

// iCamera.gluv's component along AXIS (0: x, 1: y) of the fragments whose gluv component along that axis is `g`, and whether the
// camera's plane lies behind it (t < 0, the same for all fragments) — for camera_is_axis_aligned() cameras
template <int AXIS> SF_HD float camera_along_axis(const Uniforms& u, float g, float aspect, bool& behind) {
    Frag f{};
    f.u = &u; f.aspect = aspect;
    f.gluv = AXIS == 0 ? vec2{g, 0.0f} : vec2{0.0f, g};
    f.agluv = f.gluv/vec2{aspect, 1.0f};
    const Camera c = get_camera(f);
    // get_camera's out_of_bounds is (t < 0) || |gluv.x| > iWantAspect: with gluv.x = 0 it is the first term alone
    Frag probe = f; probe.gluv = vec2{0.0f, 0.0f}; probe.agluv = vec2{0.0f, 0.0f};
    behind = get_camera(probe).out_of_bounds;
    return AXIS == 0 ? c.gluv.x : c.gluv.y;
}

}  // namespace sf
