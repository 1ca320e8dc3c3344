/*
 * ORACLE (test infrastructure, never shipped, never on the product path).
 * Pixel half: CPU restatement of the reference's GLSL (vertex/default.glsl, include/shaderflow.glsl,
 * include/camera.glsl, fragment/{default,missing,final}.glsl, examples/basic/shaders/ fragments) and of
 * the OpenGL 3.3 rules the reference leans on. PINNED against frames the reference itself rendered on Mesa llvmpipe in the
 * build container (tests/golden/mesa.npz, mesa_4k.npz; tests/test_oracle_mesa.py) and, as a second witness, against its GLSL
 * on SwiftShader (gles.npz, tests/test_oracle_gles.py) — see sfo.h; the reference itself holds no golden images.
 * Citations are file:line in /root/reference.
 *
 * Conventions fixed here (and restated independently by the HIP kernels):
 *   - pixel (i, j) of a (wr, hr) target, origin bottom-left, is shaded at its centre; the varyings
 *     are the vertex shader's formulas evaluated on agluv = 2*((i+.5)/wr, (j+.5)/hr) - 1
 *     (vertex/default.glsl:8-16; the quad's attributes are affine, shader.py:127-128)
 *   - `out vec4 fragColor` starts as vec4(0)
 *   - built-ins per sfo_math.h; every expression is evaluated left to right in binary32 with no
 *     contraction; GLSL mix/mod/smoothstep per their GLSL 3.30 §8.3 definitions
 *   - texture(): OpenGL 3.3 core §3.8.8-3.8.9 (texel addressing, wrap, bilinear), weights applied as
 *     fma(w11,t11, fma(w01,t01, fma(w10,t10, w00*t00))); unorm8 texel → c/255.0f
 *   - colour write: clamp to [0,1] (NaN → 0), *255, round half to even (OpenGL 3.3 §2.1.6/§4.1)
 */
#include "sfo.h"
#include "sfo_math.h"

#include <pthread.h>
#include <stdlib.h>

typedef struct { float x, y; } v2;
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v2 V2(float x, float y) { v2 r = {x, y}; return r; }
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v4 V4(float x, float y, float z, float w) { v4 r = {x, y, z, w}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(v3 a, float s) { return V3(a.x*s, a.y*s, a.z*s); }
static inline float v3_dot(v3 a, v3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }
static inline float v2_length(v2 a) { return sfo_sqrt(a.x*a.x + a.y*a.y); }

/* ---------------------------------------------------------------------------------------------- */
/* OpenGL 3.3 sampler */

static inline int wrap_index(int i, int size, int repeat) {
    if (repeat) { int m = i % size; return (m < 0) ? m + size : m; }      /* REPEAT: i mod size */
    return (i < 0) ? 0 : ((i >= size) ? size - 1 : i);                    /* CLAMP_TO_EDGE */
}

/* IEEE binary16 <-> binary32 (numpy float16 textures, texture.py:28-38), written out: gcc 11 has no _Float16 on x86 */
static inline float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exponent = (h >> 10) & 0x1Fu, mantissa = h & 0x3FFu, bits;
    if (exponent == 0) {
        if (mantissa == 0) bits = sign;
        else {                                                              /* subnormal: renormalise */
            int shift = 0;
            while (!(mantissa & 0x400u)) { mantissa <<= 1; shift++; }
            bits = sign | ((uint32_t)(113 - shift) << 23) | ((mantissa & 0x3FFu) << 13);
        }
    } else if (exponent == 31) bits = sign | 0x7F800000u | (mantissa << 13);
    else bits = sign | ((exponent + 112u) << 23) | (mantissa << 13);
    float f; memcpy(&f, &bits, 4); return f;
}
static inline uint16_t float_to_half(float f) {                            /* round to nearest even */
    uint32_t bits; memcpy(&bits, &f, 4);
    uint32_t sign = (bits >> 16) & 0x8000u, magnitude = bits & 0x7FFFFFFFu;
    if (magnitude >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((magnitude > 0x7F800000u) ? 0x200u : 0u));
    if (magnitude >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);        /* rounds to infinity */
    if (magnitude < 0x33000001u) return (uint16_t)sign;                     /* rounds to zero */
    int exponent = (int)(magnitude >> 23) - 127;
    uint32_t mantissa = (magnitude & 0x7FFFFFu) | 0x800000u;
    int shift = (exponent < -14) ? (-14 - exponent + 13) : 13;              /* subnormal halves lose more bits */
    uint32_t kept = mantissa >> shift, rest = mantissa & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rest > half || (rest == half && (kept & 1u))) kept++;
    if (exponent < -14) return (uint16_t)(sign | kept);                     /* kept may carry into the smallest normal: same bits */
    return (uint16_t)(sign | (((uint32_t)(exponent + 15) << 10) + (kept - 0x400u)));
}

static inline v4 fetch_texel(const sfo_texture* t, int i, int j) {
    v4 c = V4(0.0f, 0.0f, 0.0f, 1.0f);
    float* out = &c.x;
    int64_t base = ((int64_t)j*t->width + i)*t->components;
    for (int k = 0; k < t->components && k < 4; k++) {
        if (t->dtype == SFO_U8) out[k] = (float)((const uint8_t*)t->data)[base + k]/255.0f;
        else if (t->dtype == SFO_U16) out[k] = (float)((const uint16_t*)t->data)[base + k]/65535.0f;
        else if (t->dtype == SFO_F16) out[k] = half_to_float(((const uint16_t*)t->data)[base + k]);
        else out[k] = ((const float*)t->data)[base + k];
    }
    return c;
}

/* The filter of the implementation the golden frames were rendered on, as a CHECKER'S SWITCH (never the default, never in the kernels):
 * Mesa llvmpipe 23.2 samples 8-bit unorm textures through its fixed-point path — texel coordinates in 24.8 fixed point,
 * fx = rint(s*N*256) - 128, weight = fx & 255 (8 fractional bits, the coordinate ROUNDED, not truncated), i0 = fx >> 8; the wrap mode
 * applied to i0 and i0+1; one lerp along x on the 8-bit values, a + ((w*(b - a) + 128) >> 8), ROUNDED BACK TO 8 BITS, then the same
 * lerp along y — so a filtered unorm8 texel is always k/255. OpenGL allows it (>= 4 subtexel bits, section 3.8.11). Measured, not
 * read: tests/golden/make_golden_filter.py renders probe textures through the reference on llvmpipe into a float32 target; this
 * model reproduces all 2.8 M filtered values of that fixture bit for bit (tests/test_oracle_mesa.py). With the switch on, the
 * oracle's frames meet the llvmpipe goldens at <= 1 LSB where the spec-precision filter leaves up to 1.3 % of the values 2 off. */
static int g_llvmpipe_filter = 0;
void sfo_set_llvmpipe_filter(int on) { g_llvmpipe_filter = on; }

static inline int fixed_wrap(int i, int n, int repeat) {
    if (repeat) { i %= n; return i < 0 ? i + n : i; }
    return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}
static inline int fixed_lerp(int w, int a, int b) { return (a + ((w*(b - a) + 128) >> 8)) & 255; }

static v4 sample_llvmpipe_u8(const sfo_texture* t, v2 uv) {
    const int fx = (int)lrintf(uv.x*(float)t->width*256.0f) - 128, fy = (int)lrintf(uv.y*(float)t->height*256.0f) - 128;
    const int wx = fx & 255, wy = fy & 255;
    const int i0 = fixed_wrap(fx >> 8, t->width, t->repeat_x), i1 = fixed_wrap((fx >> 8) + 1, t->width, t->repeat_x);
    const int j0 = fixed_wrap(fy >> 8, t->height, t->repeat_y), j1 = fixed_wrap((fy >> 8) + 1, t->height, t->repeat_y);
    const uint8_t* data = (const uint8_t*)t->data;
    v4 c = V4(0.0f, 0.0f, 0.0f, 1.0f);
    float* out = &c.x;
    for (int k = 0; k < t->components && k < 4; k++) {
        const int t00 = data[((int64_t)j0*t->width + i0)*t->components + k], t10 = data[((int64_t)j0*t->width + i1)*t->components + k];
        const int t01 = data[((int64_t)j1*t->width + i0)*t->components + k], t11 = data[((int64_t)j1*t->width + i1)*t->components + k];
        out[k] = (float)fixed_lerp(wy, fixed_lerp(wx, t00, t10), fixed_lerp(wx, t01, t11))/255.0f;
    }
    return c;
}

/* ---- mipmapped sampling: OpenGL 3.3 core section 3.8.11 (texture.py:131-137, 277-278) ----------------------------------------------
 * The level of detail needs the derivatives of the coordinate across the pixel's 2x2 quad. The fragments here are plain functions of
 * one pixel, so render_rows shades a pixel of a frame with a mipmapped texture THREE times: its horizontal and its vertical quad
 * neighbour first, RECORDING the texel-space coordinate of every mipmapped texture() call in order, then the pixel itself, where call
 * number k finds its neighbours' coordinates of call number k (uniform control flow across a quad is GLSL's own condition for
 * implicit derivatives). rho = the longer footprint axis in level-0 texels, lambda = log2(rho), clamped to the chain; lambda <= 0:
 * the magnification filter on level 0; LINEAR_MIPMAP_LINEAR: levels floor(lambda), +1 blended by frac(lambda);
 * NEAREST_MIPMAP_NEAREST: nearest texel of level ceil(lambda + 0.5) - 1.
 * Under the llvmpipe switch: lambda = log2(rho^2)/2 with log2 taken as exponent + (mantissa - 1), the blend weight trunc(frac*256) and
 * a + ((w*(b - a) + 128) >> 8) on the two levels' 8-bit results (unorm8; measured like the filter itself, tests/golden/mip.npz). */
static inline uint8_t to_unorm8(float c);
enum { LOD_OFF = 0, LOD_RECORD_X = 1, LOD_RECORD_Y = 2, LOD_REPLAY = 3, LOD_CALLS = 256 };
static __thread struct { int mode, count; float ux[LOD_CALLS], vx[LOD_CALLS], uy[LOD_CALLS], vy[LOD_CALLS]; } g_lod;

int sfo_mip_levels(int width, int height) {
    int levels = 1;
    while (width > 1 || height > 1) { width = width > 1 ? width >> 1 : 1; height = height > 1 ? height >> 1 : 1; levels++; }
    return levels;
}
static inline int64_t texel_size(const sfo_texture* t) { return (int64_t)t->components*(t->dtype == SFO_U8 ? 1 : (t->dtype == SFO_F32 ? 4 : 2)); }
int64_t sfo_mip_offset(const sfo_texture* t, int level) {
    int64_t offset = 0;
    int w = t->width, h = t->height;
    for (int l = 1; l < level; l++) { w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1; offset += (w*(int64_t)h*texel_size(t) + 15) & ~(int64_t)15; }
    return offset;
}
static sfo_texture mip_level(const sfo_texture* t, int k) {
    sfo_texture v = *t;
    v.levels = 0; v.mips = NULL;
    if (k <= 0 || t->levels <= 1 || !t->mips) return v;
    if (k > t->levels - 1) k = t->levels - 1;
    for (int l = 0; l < k; l++) { v.width = v.width > 1 ? v.width >> 1 : 1; v.height = v.height > 1 ? v.height >> 1 : 1; }
    v.data = (const char*)t->mips + sfo_mip_offset(t, k);
    return v;
}
static v4 sample(const sfo_texture* t, v2 uv);
void sfo_build_mipmaps(const sfo_texture* t, void* mips) {
    sfo_texture whole = *t;
    whole.levels = sfo_mip_levels(t->width, t->height); whole.mips = mips;
    for (int l = 1; l < whole.levels; l++) {
        sfo_texture above = mip_level(&whole, l - 1), here = mip_level(&whole, l);
        above.filter = SFO_LINEAR; above.repeat_x = 0; above.repeat_y = 0;
        for (int j = 0; j < here.height; j++) {
            for (int i = 0; i < here.width; i++) {
                v4 c;
                if (g_llvmpipe_filter && t->dtype == SFO_U8) {        /* the blit's own coordinates: normalised, as a varying */
                    c = sample(&above, V2(((float)i + 0.5f)/(float)here.width, ((float)j + 0.5f)/(float)here.height));
                } else {                                              /* exact texel coordinates: an exact halving weighs 1/2, 1/2 */
                    const double ub = ((double)i + 0.5)*(double)above.width/(double)here.width - 0.5, vb = ((double)j + 0.5)*(double)above.height/(double)here.height - 0.5;
                    const double fu = floor(ub), fv = floor(vb);
                    const float ax = (float)(ub - fu), ay = (float)(vb - fv);
                    const int i0 = wrap_index((int)fu, above.width, 0), i1 = wrap_index((int)fu + 1, above.width, 0);
                    const int j0 = wrap_index((int)fv, above.height, 0), j1 = wrap_index((int)fv + 1, above.height, 0);
                    const v4 t00 = fetch_texel(&above, i0, j0), t10 = fetch_texel(&above, i1, j0), t01 = fetch_texel(&above, i0, j1), t11 = fetch_texel(&above, i1, j1);
                    const float nx = 1.0f - ax, ny = 1.0f - ay, w00 = nx*ny, w10 = ax*ny, w01 = nx*ay, w11 = ax*ay;
                    c.x = fmaf(w11, t11.x, fmaf(w01, t01.x, fmaf(w10, t10.x, w00*t00.x)));
                    c.y = fmaf(w11, t11.y, fmaf(w01, t01.y, fmaf(w10, t10.y, w00*t00.y)));
                    c.z = fmaf(w11, t11.z, fmaf(w01, t01.z, fmaf(w10, t10.z, w00*t00.z)));
                    c.w = fmaf(w11, t11.w, fmaf(w01, t01.w, fmaf(w10, t10.w, w00*t00.w)));
                }
                const float channel[4] = {c.x, c.y, c.z, c.w};
                const int64_t at = ((int64_t)j*here.width + i)*t->components;
                for (int k = 0; k < t->components; k++) {
                    if (t->dtype == SFO_U8) ((uint8_t*)here.data)[at + k] = to_unorm8(channel[k]);
                    else if (t->dtype == SFO_F32) ((float*)here.data)[at + k] = channel[k];
                    else if (t->dtype == SFO_F16) ((uint16_t*)here.data)[at + k] = float_to_half(channel[k]);
                    else { float q = channel[k] > 0.0f ? channel[k] : 0.0f; q = q < 1.0f ? q : 1.0f; ((uint16_t*)here.data)[at + k] = (uint16_t)rintf(q*65535.0f); }
                }
            }
        }
    }
}

static v4 sample_level(const sfo_texture* t, v2 uv, int level, int filter) {
    sfo_texture v = mip_level(t, level);
    v.filter = filter;
    return sample(&v, uv);
}
static v4 sample_mipmapped(const sfo_texture* t, v2 uv) {
    const int linear = (t->filter == SFO_LINEAR_MIPMAP), magnify = linear ? SFO_LINEAR : SFO_NEAREST;
    const float u = uv.x*(float)t->width, v = uv.y*(float)t->height;
    const int k = g_lod.count++;
    if (g_lod.mode == LOD_RECORD_X) { if (k < LOD_CALLS) { g_lod.ux[k] = u; g_lod.vx[k] = v; } return sample_level(t, uv, 0, magnify); }
    if (g_lod.mode == LOD_RECORD_Y) { if (k < LOD_CALLS) { g_lod.uy[k] = u; g_lod.vy[k] = v; } return sample_level(t, uv, 0, magnify); }
    if (g_lod.mode != LOD_REPLAY || k >= LOD_CALLS || t->levels <= 1) return sample_level(t, uv, 0, magnify);
    const float dudx = u - g_lod.ux[k], dvdx = v - g_lod.vx[k], dudy = u - g_lod.uy[k], dvdy = v - g_lod.vy[k];
    const float along_x = dudx*dudx + dvdx*dvdx, along_y = dudy*dudy + dvdy*dvdy;
    const float rho2 = along_x > along_y ? along_x : along_y;
    float lambda;
    if (g_llvmpipe_filter) {
        int exponent = 0;
        const float mantissa = frexpf(rho2, &exponent)*2.0f;         /* rho2 = mantissa * 2^(exponent-1), mantissa in [1, 2) */
        lambda = rho2 > 0.0f ? 0.5f*((float)(exponent - 1) + (mantissa - 1.0f)) : -INFINITY;
    } else {
        lambda = 0.5f*sfo_log2(rho2);
    }
    const float top = (float)(t->levels - 1);
    if (lambda > top) lambda = top;
    if (!(lambda > 0.0f)) return sample_level(t, uv, 0, magnify);
    if (!linear) return sample_level(t, uv, (int)ceilf(lambda + 0.5f) - 1, SFO_NEAREST);
    const float below = floorf(lambda), f = lambda - below;
    const int d1 = (int)below, d2 = d1 + 1 < t->levels ? d1 + 1 : t->levels - 1;
    const v4 a = sample_level(t, uv, d1, SFO_LINEAR);
    if (d2 == d1 || f == 0.0f) return a;
    const v4 b = sample_level(t, uv, d2, SFO_LINEAR);
    v4 r;
    if (g_llvmpipe_filter && t->dtype == SFO_U8) {
        const int w = (int)(f*256.0f);
        const float* pa = &a.x; const float* pb = &b.x; float* pr = &r.x;
        for (int c = 0; c < 4; c++) pr[c] = (float)fixed_lerp(w, (int)lrintf(pa[c]*255.0f), (int)lrintf(pb[c]*255.0f))/255.0f;
        if (t->components < 4) r.w = 1.0f;
        return r;
    }
    r.x = fmaf(f, b.x - a.x, a.x); r.y = fmaf(f, b.y - a.y, a.y); r.z = fmaf(f, b.z - a.z, a.z); r.w = fmaf(f, b.w - a.w, a.w);
    return r;
}

static v4 sample(const sfo_texture* t, v2 uv) {
    if (t->filter >= SFO_LINEAR_MIPMAP) return sample_mipmapped(t, uv);
    float u = uv.x*(float)t->width;
    float v = uv.y*(float)t->height;
    if (t->filter == SFO_NEAREST) {
        int i = wrap_index((int)floorf(u), t->width, t->repeat_x);
        int j = wrap_index((int)floorf(v), t->height, t->repeat_y);
        return fetch_texel(t, i, j);
    }
    if (g_llvmpipe_filter && t->dtype == SFO_U8) return sample_llvmpipe_u8(t, uv);
    float ub = u - 0.5f, vb = v - 0.5f;
    float fu = floorf(ub), fv = floorf(vb);
    float a = ub - fu, b = vb - fv;
    int i0 = wrap_index((int)fu, t->width, t->repeat_x), i1 = wrap_index((int)fu + 1, t->width, t->repeat_x);
    int j0 = wrap_index((int)fv, t->height, t->repeat_y), j1 = wrap_index((int)fv + 1, t->height, t->repeat_y);
    v4 t00 = fetch_texel(t, i0, j0), t10 = fetch_texel(t, i1, j0);
    v4 t01 = fetch_texel(t, i0, j1), t11 = fetch_texel(t, i1, j1);
    float na = 1.0f - a, nb = 1.0f - b;
    float w00 = na*nb, w10 = a*nb, w01 = na*b, w11 = a*b;
    v4 r;
    r.x = fmaf(w11, t11.x, fmaf(w01, t01.x, fmaf(w10, t10.x, w00*t00.x)));
    r.y = fmaf(w11, t11.y, fmaf(w01, t01.y, fmaf(w10, t10.y, w00*t00.y)));
    r.z = fmaf(w11, t11.z, fmaf(w01, t01.z, fmaf(w10, t10.z, w00*t00.z)));
    r.w = fmaf(w11, t11.w, fmaf(w01, t01.w, fmaf(w10, t10.w, w00*t00.w)));
    return r;
}

/* The optional half of the encoder hand-off: RGB8 → planar yuv420p, the arithmetic the product DEFINES for it
 * (shaderflow_amd/csrc/capi.hip k_rgb_to_yuv420; there is no reference implementation to restate — the reference leaves the conversion
 * to ffmpeg's swscale, and no ffmpeg binary exists here). matrix 0: BT.601 limited range, 1: BT.709 limited range; 8-bit integer
 * coefficients, arithmetic shifts, chroma from the rounded mean of the 2x2 block's R, G, B. */
void sfo_rgb_to_yuv420(const uint8_t* rgb, int w, int h, int matrix, uint8_t* yuv) {
    static const int M[2][9] = {{66, 129, 25, -38, -74, 112, 112, -94, -18}, {47, 157, 16, -26, -86, 112, 112, -102, -10}};
    const int* m = M[matrix == 1];
    uint8_t* u_plane = yuv + (int64_t)w*h;
    uint8_t* v_plane = u_plane + (int64_t)(w/2)*(h/2);
    for (int by = 0; by < h/2; by++) {
        for (int bx = 0; bx < w/2; bx++) {
            int sr = 0, sg = 0, sb = 0;
            for (int y = 0; y < 2; y++) {
                for (int x = 0; x < 2; x++) {
                    const uint8_t* p = rgb + ((int64_t)(2*by + y)*w + 2*bx + x)*3;
                    sr += p[0]; sg += p[1]; sb += p[2];
                    yuv[(int64_t)(2*by + y)*w + 2*bx + x] = (uint8_t)(((m[0]*p[0] + m[1]*p[1] + m[2]*p[2] + 128) >> 8) + 16);
                }
            }
            const int r = (sr + 2) >> 2, g = (sg + 2) >> 2, b = (sb + 2) >> 2;
            u_plane[(int64_t)by*(w/2) + bx] = (uint8_t)(((m[3]*r + m[4]*g + m[5]*b + 128) >> 8) + 128);
            v_plane[(int64_t)by*(w/2) + bx] = (uint8_t)(((m[6]*r + m[7]*g + m[8]*b + 128) >> 8) + 128);
        }
    }
}

void sfo_sample_quad(const sfo_texture* t, float s, float tt, float s_right, float t_right, float s_above, float t_above, float rgba[4]) {
    g_lod.mode = LOD_REPLAY; g_lod.count = 0;
    g_lod.ux[0] = s_right*(float)t->width; g_lod.vx[0] = t_right*(float)t->height;
    g_lod.uy[0] = s_above*(float)t->width; g_lod.vy[0] = t_above*(float)t->height;
    v4 c = sample(t, V2(s, tt));
    g_lod.mode = LOD_OFF;
    rgba[0] = c.x; rgba[1] = c.y; rgba[2] = c.z; rgba[3] = c.w;
}

void sfo_sample(const sfo_texture* t, float s, float tt, float rgba[4]) {
    v4 c = sample(t, V2(s, tt));
    rgba[0] = c.x; rgba[1] = c.y; rgba[2] = c.z; rgba[3] = c.w;
}

static inline uint8_t to_unorm8(float c) {
    c = (c > 0.0f) ? c : 0.0f;            /* NaN → 0 */
    c = (c < 1.0f) ? c : 1.0f;
    return (uint8_t)rintf(c*255.0f);
}

/* ---------------------------------------------------------------------------------------------- */
/* Varyings and prelude */

typedef struct {
    const sfo_uniforms* u;
    const sfo_texture* tex;
    v2 agluv, gluv, astuv, stuv, stxy, glxy, fragCoord;
    float aspect;                         /* iAspectRatio, shaderflow.glsl:16 */
} frag_in;

/* camera.glsl:15-51 */
typedef struct {
    v3 position, up, right, forward, backward, origin, target, plane_point, plane_normal;
    float orbital, dolly, separation, focal_length, isometric, zoom;
    int projection;
    v2 gluv, agluv, stuv, astuv, glxy, stxy;
    int out_of_bounds;
} camera_t;

static inline v2 gluv2stuv(v2 g) { return V2((g.x + 1.0f)/2.0f, (g.y + 1.0f)/2.0f); }        /* shaderflow.glsl:95 */
static inline v2 stuv2gluv(v2 s) { return V2((s.x*2.0f) - 1.0f, (s.y*2.0f) - 1.0f); }        /* shaderflow.glsl:91 */

static void make_varyings(frag_in* f, int i, int j, int wr, int hr) {
    const sfo_uniforms* u = f->u;
    f->aspect = u->iResolution[0]/u->iResolution[1];
    v2 astuv0 = V2(((float)i + 0.5f)/(float)wr, ((float)j + 0.5f)/(float)hr);
    f->agluv = V2(astuv0.x*2.0f - 1.0f, astuv0.y*2.0f - 1.0f);
    f->gluv = V2(f->agluv.x*f->aspect, f->agluv.y*1.0f);                 /* shaderflow.glsl:99 */
    f->astuv = gluv2stuv(f->agluv);                                       /* vertex/default.glsl:10 */
    f->stuv = gluv2stuv(f->gluv);                                         /* :11 */
    f->stxy = V2(u->iResolution[0]*f->astuv.x + 1.0f, u->iResolution[1]*f->astuv.y + 1.0f);   /* :14 */
    f->glxy = V2(f->stxy.x - u->iResolution[0]/2.0f, f->stxy.y - u->iResolution[1]/2.0f);     /* :15 */
    f->fragCoord = f->stxy;                                               /* :16 */
}

/* shaderflow.glsl:75-77; GLSL mat2(a,b,c,d) is column-major: m*v = (a*x + c*y, b*x + d*y) */
static inline v2 rotate2d_mul(float angle, v2 p) {
    float c = sfo_cos(angle), s = sfo_sin(angle);
    return V2(c*p.x + s*p.y, (-s)*p.x + c*p.y);
}
/* shaderflow.glsl:361-363 */
static inline v2 zoom_anchor(v2 uv, float zoom, v2 anchor) {
    float z2 = zoom*zoom;
    return V2((uv.x - anchor.x)*z2 + anchor.x, (uv.y - anchor.y)*z2 + anchor.y);
}
/* shaderflow.glsl:165-169: textureSize, scale = (res.y/res.x, 1), texture(image, gluv2stuv(gluv*scale)) */
static inline v4 gtexture(const sfo_texture* t, v2 gluv) {
    v2 scale = V2((float)t->height/(float)t->width, 1.0f);
    return sample(t, gluv2stuv(V2(gluv.x*scale.x, gluv.y*scale.y)));
}
static inline v4 stexture(const sfo_texture* t, v2 stuv) { return gtexture(t, stuv2gluv(stuv)); }   /* :198-200 */
static inline float atan1n(v2 p) { return sfo_atan2(p.y, p.x)/SFO_PI; }                              /* :378-380 */
static inline float atan2_0_tau(float y, float x) {                                                  /* :382-388 */
    if (y < 0.0f) return SFO_TAU - sfo_atan2(-y, x);
    return sfo_atan2(y, x);
}
/* shaderflow.glsl:406-425 */
static v3 hsv2rgb(float h, float s, float v) {
    h = sfo_mod(h, SFO_TAU);
    float c = v*s;
    float x = c*(1.0f - sfo_abs(sfo_mod(h/(SFO_PI/3.0f), 2.0f) - 1.0f));
    float m = v - c;
    v3 rgb;
    switch (sfo_to_int(floorf(6.0f*(h/(2.0f*SFO_PI))))) {
        case 0: rgb = V3(c, x, 0.0f); break;
        case 1: rgb = V3(x, c, 0.0f); break;
        case 2: rgb = V3(0.0f, c, x); break;
        case 3: rgb = V3(0.0f, x, c); break;
        case 4: rgb = V3(x, 0.0f, c); break;
        case 5: rgb = V3(c, 0.0f, x); break;
        default: rgb = V3(0.0f, 0.0f, 0.0f);
    }
    return V3(rgb.x + m, rgb.y + m, rgb.z + m);
}

/* camera.glsl:53-71 */
static inline v3 camera_rectangle(const camera_t* c, v2 gluv, float size) {
    v3 a = v3_scale(c->right, gluv.x), b = v3_scale(c->up, gluv.y);
    return v3_scale(v3_add(a, b), size);
}
static inline v3 camera_ray_origin(const camera_t* c, v2 gluv) {
    v3 r = v3_add(c->position, camera_rectangle(c, gluv, c->zoom*c->isometric));
    r = v3_add(r, v3_scale(c->backward, c->orbital));
    return v3_add(r, v3_scale(c->backward, c->dolly));
}
static inline v3 camera_ray_target(const camera_t* c, v2 gluv) {
    v3 r = v3_add(c->position, camera_rectangle(c, gluv, c->zoom));
    r = v3_add(r, v3_scale(c->backward, c->orbital));
    return v3_add(r, v3_scale(c->forward, c->focal_length));
}
/* shaderflow.glsl:81-83: mix(dot(axis,v)*axis, v, cos) + cross(axis,v)*sin */
static v3 rotate3d(v3 vector, v3 axis, float angle) {
    float c = sfo_cos(angle), s = sfo_sin(angle);
    float d = v3_dot(axis, vector);
    v3 pa = v3_scale(axis, d);
    v3 m = V3(sfo_mix(pa.x, vector.x, c), sfo_mix(pa.y, vector.y, c), sfo_mix(pa.z, vector.z, c));
    v3 cr = V3(axis.y*vector.z - vector.y*axis.z, axis.z*vector.x - vector.z*axis.x, axis.x*vector.y - vector.x*axis.y);
    return v3_add(m, v3_scale(cr, s));
}
static inline float sfo_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

/* camera.glsl:132-155 (GetCamera) → :93-130 (CameraProject) → :73-91 (CameraRay2D) */
static camera_t get_camera(const frag_in* f) {
    const sfo_uniforms* u = f->u;
    camera_t c;
    c.plane_point = V3(0.0f, 0.0f, 1.0f);
    c.plane_normal = V3(0.0f, 0.0f, 1.0f);
    c.projection = u->iCameraProjection;
    c.position = V3(u->iCameraPosition[0], u->iCameraPosition[1], u->iCameraPosition[2]);
    c.orbital = u->iCameraOrbital;
    c.dolly = u->iCameraDolly;
    c.up = V3(u->iCameraUpward[0], u->iCameraUpward[1], u->iCameraUpward[2]);
    c.right = V3(u->iCameraRight[0], u->iCameraRight[1], u->iCameraRight[2]);
    c.forward = V3(u->iCameraForward[0], u->iCameraForward[1], u->iCameraForward[2]);
    c.backward = v3_scale(c.forward, -1.0f);
    c.isometric = u->iCameraIsometric;
    c.focal_length = u->iCameraFocalLength;
    c.zoom = u->iCameraZoom;
    c.separation = u->iCameraSeparation;
    c.out_of_bounds = 0;

    if (c.projection == 0) {                                              /* Perspective, :96-98 */
        c.origin = camera_ray_origin(&c, f->gluv);
        c.target = camera_ray_target(&c, f->gluv);
    } else if (c.projection == 1) {                                       /* Stereoscopic, :101-110 */
        float sg = sfo_sign(f->agluv.x);
        v2 g = V2(f->gluv.x - sg*(f->aspect/2.0f), f->gluv.y - sg*0.0f);
        c.position = v3_add(c.position, v3_scale(c.right, sg*c.separation));
        c.origin = camera_ray_origin(&c, g);
        c.target = camera_ray_target(&c, g);
    } else {                                                              /* Equirectangular, :113-126 */
        float inclination = c.zoom*(SFO_PI*f->agluv.y/2.0f);
        float azimuth = c.zoom*(SFO_PI*f->agluv.x/1.0f);
        v3 target = c.forward;
        target = rotate3d(target, c.right, -inclination);
        target = rotate3d(target, c.up, azimuth);
        c.origin = c.position;
        c.target = v3_add(c.position, target);
    }

    /* CameraRay2D, :73-91 */
    float num = v3_dot(v3_sub(c.plane_point, c.origin), c.plane_normal);
    float den = v3_dot(v3_sub(c.target, c.origin), c.plane_normal);
    float t = num/den;
    c.out_of_bounds = (t < 0.0f) || (sfo_abs(f->gluv.x) > u->iWantAspect);
    v3 hit = v3_add(c.origin, v3_scale(v3_sub(c.target, c.origin), t));
    c.gluv = V2(hit.x, hit.y);
    c.agluv = V2(c.gluv.x/f->aspect, c.gluv.y/1.0f);
    c.stuv = V2((c.gluv.x + 1.0f)/2.0f, (c.gluv.y + 1.0f)/2.0f);
    c.astuv = V2((c.agluv.x + 1.0f)/2.0f, (c.agluv.y + 1.0f)/2.0f);
    c.stxy = V2(u->iResolution[0]*c.astuv.x, u->iResolution[1]*c.astuv.y);
    c.glxy = V2(c.stxy.x - u->iResolution[0]/2.0f, c.stxy.y - u->iResolution[1]/2.0f);
    return c;
}

/* ---------------------------------------------------------------------------------------------- */
/* Fragments */

/* fragment/default.glsl:1-48 */
static v3 default_grid(v2 uv, float grid) {                               /* :4-8 */
    if (sfo_mod(floorf(uv.x*grid/2.0f) + floorf(uv.y*grid/2.0f), 2.0f) > 0.5f) return V3(0.22f, 0.22f, 0.22f);
    return V3(0.20f, 0.20f, 0.20f);
}
static v4 frag_default(const frag_in* f) {
    const sfo_uniforms* u = f->u;
    camera_t cam = get_camera(f);
    v4 col = V4(0.0f, 0.0f, 0.0f, 0.0f);
    v2 uv = cam.gluv;
    if (cam.out_of_bounds) return V4(0.15f, 0.15f, 0.15f, 1.0f);          /* :15-17 */
    float angle = atan2_0_tau(uv.y, uv.x);                                /* :21 */
    v3 hsv = hsv2rgb(angle + (2.0f*SFO_TAU*u->iTau) - (SFO_PI/4.0f), 1.0f, 1.0f);
    v3 color = V3(0.3f + hsv.x, 0.3f + hsv.y, 0.3f + hsv.z);             /* :24 */
    float circle = (1.333f*v2_length(uv) - 1.0f);                         /* :27 */
    float width = 2.0f*sfo_abs(1.0f/(circle*circle))*1e-4f;               /* :28 */
    if (circle < 0.0f) { col.x += 0.18f; col.y += 0.18f; col.z += 0.18f; }   /* :31-32 */
    else { v3 g = default_grid(uv, 8.0f); col.x += g.x; col.y += g.y; col.z += g.z; }   /* :34 (LOGO false) */
    col.x += (width*color.x); col.y += (width*color.y); col.z += (width*color.z);        /* :38 */
    col.w = 1.0f;
    v2 away = V2(f->astuv.x*(1.0f - f->astuv.y), f->astuv.y*(1.0f - f->astuv.x));       /* :42 */
    float linear = 50.0f*(away.x*away.y);                                 /* :43 */
    float vig = sfo_clamp(sfo_pow(linear, 0.1f), 0.0f, 1.0f);             /* :44 */
    col.x *= vig; col.y *= vig; col.z *= vig;
    return col;
}

/* fragment/missing.glsl:4-22 */
static v4 frag_missing(const frag_in* f) {
    v4 col = V4(0.0f, 0.0f, 0.0f, 0.0f);
    v2 uv = V2(f->stuv.x + f->u->iTime/64.0f, f->stuv.y + f->u->iTime/64.0f);
    float size = 8.0f;
    for (int x = -5; x < 5; x++) {
        for (int y = -5; y < 5; y++) {
            v2 block = V2(floorf(size*uv.x), floorf(size*uv.y));
            if (sfo_mod(block.x + block.y, 2.0f) == 0.0f) {
                col.x += 1.0f/25.0f; col.y += 0.0f/25.0f; col.z += 1.0f/25.0f;
            }
        }
    }
    col.w = 0.2f;
    return col;
}

/* examples/basic/shaders/visualizer.frag:6-74 */
static v4 frag_visualizer(const frag_in* f) {
    const sfo_uniforms* u = f->u;
    const sfo_texture* background = &f->tex[SFO_TEX_BACKGROUND];
    camera_t cam = get_camera(f);
    v2 uv = cam.gluv;
    v3 space = V3(1.0f/255.0f, 11.0f/255.0f, 26.0f/255.0f);               /* :9 */
    v4 col = V4(0.0f, 0.0f, 0.0f, 0.0f);
    if (cam.out_of_bounds) { col.x = space.x; col.y = space.y; col.z = space.z; return col; }   /* :11-14 */

    /* :17-19 */
    v2 bg = zoom_anchor(gluv2stuv(uv), 0.95f + 0.01f*sfo_sin(u->iTime) - 0.02f*u->iAudioVolume - 0.03f, V2(0.5f, 0.5f));
    bg.x += 0.005f*sfo_cos(u->iTime*3.25135f);
    bg.y += 0.005f*sfo_sin(u->iTime*1.153469f);
    col = stexture(background, bg);

    {   /* :21-33; float loop counters evaluated in binary32 (9 directions x 10 steps) */
        float intensity = 0.01f*sfo_clamp(sfo_pow(u->iAudioVolume, 2.5f), 0.0f, 0.3f);
        float quality = 10.0f;
        float directions = 8.0f;
        v4 color = col;
        for (float angle = 0.0f; angle < SFO_TAU; angle += SFO_TAU/directions) {
            for (float walk = 1.0f/quality; walk <= 1.001f; walk += 1.0f/quality) {
                v2 disp = V2(sfo_cos(angle)*walk*intensity, sfo_sin(angle)*walk*intensity);
                v4 s = stexture(background, V2(bg.x + disp.x, bg.y + disp.y));
                color.x += s.x; color.y += s.y; color.z += s.z; color.w += s.w;
            }
        }
        float div = quality*directions;
        col = V4(color.x/div, color.y/div, color.z/div, color.w/div);
    }

    {   /* :36 */
        float k = 1.0f + 5.0f*u->iAudioSTD*sfo_pow(sfo_clamp(v2_length(f->agluv) - 0.3f, 0.0f, 1.0f), 6.0f);
        col.x *= k; col.y *= k; col.z *= k; col.w *= k;
    }

    /* :39-41 */
    v2 music_uv = rotate2d_mul(-SFO_PI/2.0f, uv);
    float shrink = 1.0f - 0.4f*sfo_pow(sfo_abs(u->iAudioVolume), 0.5f);
    music_uv.x *= shrink; music_uv.y *= shrink;
    float radius = 0.17f;

    /* :44-46 */
    float circle = sfo_abs(atan1n(music_uv));
    v4 spec = sample(&f->tex[SFO_TEX_SPECTROGRAM], V2(0.0f, circle));
    v2 freq = V2(sfo_sqrt(spec.x/1000.0f), sfo_sqrt(spec.y/1000.0f));
    float gain = 0.05f + 3.0f*sfo_smoothstep(0.0f, 2.0f, circle);
    freq.x *= gain; freq.y *= gain;

    /* :49-60 */
    float len = v2_length(music_uv);
    if (len < radius) {
        col.x *= 0.5f; col.y *= 0.5f; col.z *= 0.5f;
    } else {
        float bar = (music_uv.y < 0.0f) ? freq.x : freq.y;
        float r = radius + 0.5f*bar;
        if (len < r) {
            float t = sfo_smoothstep(0.0f, 1.0f, 0.5f + bar);
            col.x = sfo_mix(col.x, 1.0f, t); col.y = sfo_mix(col.y, 1.0f, t); col.z = sfo_mix(col.z, 1.0f, t);
        } else {
            float k = sfo_pow((len - r)*0.5f, 0.05f);
            col.x *= k; col.y *= k; col.z *= k;
        }
    }

    {   /* :62 */
        float t = sfo_smoothstep(0.0f, 1.0f, v2_length(uv)/20.0f);
        col.x = sfo_mix(col.x, space.x, t); col.y = sfo_mix(col.y, space.y, t); col.z = sfo_mix(col.z, space.z, t);
    }

    {   /* :65-67 */
        v2 vig = V2(f->astuv.x*(1.0f - f->astuv.y), f->astuv.y*(1.0f - f->astuv.x));
        float k = sfo_pow(vig.x*vig.y*20.0f, 0.1f + 0.15f*u->iAudioVolume);
        col.x *= k; col.y *= k; col.z *= k;
        col.w = 1.0f;
    }

    {   /* :71-73 */
        v4 w = sample(&f->tex[SFO_TEX_WAVEFORM], V2(f->astuv.x, 0.0f));
        v2 wave = V2(0.2f*w.x, 0.2f*w.y);
        if (1.0f - f->gluv.y < wave.x) { col.x *= 0.8f; col.y *= 0.8f; col.z *= 0.8f; col.w *= 0.8f; }
        if (1.0f + f->gluv.y < wave.y) { col.x *= 0.8f; col.y *= 0.8f; col.z *= 0.8f; col.w *= 0.8f; }
    }
    return col;
}

/* examples/basic/shaders/bars.frag:5-22 */
static v4 frag_bars(const frag_in* f) {
    v4 col = V4(0.0f, 0.0f, 0.0f, 0.0f);
    v4 s = sample(&f->tex[SFO_TEX_SPECTROGRAM], V2(f->astuv.y, f->astuv.x));
    v2 intensity = V2(sfo_sqrt(s.x)/120.0f, sfo_sqrt(s.y)/120.0f);
    if (f->astuv.y < intensity.x) { col.x += 1.0f; col.y += 0.0f; col.z += 0.0f; }
    if (f->astuv.y < intensity.y) { col.x += 0.0f; col.y += 1.0f; col.z += 0.0f; }
    if (f->astuv.y < (intensity.y + intensity.x)/2.0f) { col.x += 0.0f; col.y += 0.0f; col.z += 1.0f; }
    col.z += 0.4f*(intensity.x + intensity.y)*(1.0f - f->astuv.y);
    col.w = 1.0f;
    return col;
}

/* examples/basic/shaders/waveform.frag:5-19 */
static v4 frag_waveform(const frag_in* f) {
    v4 w = sample(&f->tex[SFO_TEX_WAVEFORM], V2(f->astuv.x, 0.0f));
    v4 col = V4(0.2f, 0.2f, 0.2f, 1.0f);
    float ay = sfo_abs(f->gluv.y);
    if (ay < w.x) col.x = 1.0f;
    if (ay < w.y) col.y = 1.0f;
    if (ay < (w.x + w.y)/2.0f) col.z = 1.0f;
    return col;
}

/* examples/basic/demo.py:74-79 and :83-89 (MultiShader) */
static v4 frag_multi_child(const frag_in* f) { return V4(0.0f, 1.0f - f->stuv.x, 0.0f, 1.0f); }
static v4 frag_multi_main(const frag_in* f) {
    v4 c = sample(&f->tex[SFO_TEX_CHILD], f->astuv);
    return V4(f->stuv.x + c.x, 0.0f + c.y, 0.0f + c.z, 1.0f);
}

/* examples/basic/shaders/shadertoy.frag:62-66 */
static v4 frag_shadertoy(const frag_in* f) {
    float t = f->u->iTime;
    return V4(0.5f + 0.5f*sfo_cos(t + f->stuv.x + 0.0f),
              0.5f + 0.5f*sfo_cos(t + f->stuv.y + 2.0f),
              0.5f + 0.5f*sfo_cos(t + f->stuv.x + 4.0f), 1.0f);
}

/* examples/basic/demo.py:121-126 (Dynamics): user[0] = iShaderDynamics */
static v4 frag_dynamics(const frag_in* f) {
    v2 uv = zoom_anchor(f->stuv, 0.85f + 0.1f*f->u->user[0], V2(0.5f, 0.5f));
    return stexture(&f->tex[SFO_TEX_BACKGROUND], uv);
}

/* examples/basic/demo.py:149-153 (Audio) */
static v4 frag_audio(const frag_in* f) {
    float v = f->u->iAudioVolume;
    return V4(v, v, v, 1.0f);
}

/* ---------------------------------------------------------------------------------------------- */
/* Multipass / temporal fragments: textures[SFO_TEX_HISTORY + t] is `<name>{t}x0`, t frames back
 * (texture.py:346-347, 380-381; shader.py:399-405 renders row 0 layer by layer, then rolls) */

/* texelFetch: out-of-range is undefined in GL 3.3; robust-access drivers return zeros — the rule here */
static v4 texel_fetch(const sfo_texture* t, int i, int j) {
    if (i < 0 || j < 0 || i >= t->width || j >= t->height) return V4(0.0f, 0.0f, 0.0f, 0.0f);
    return fetch_texel(t, i, j);
}

static inline v4 v4_axpy(v4 acc, v4 x, float a) { return V4(acc.x + x.x*a, acc.y + x.y*a, acc.z + x.z*a, acc.w + x.w*a); }

/* examples/basic/shaders/multipass.frag:10-26 */
static v4 multipass_blur(const sfo_texture* image, v2 stuv, float radius, int directions, int steps) {
    v4 color = V4(0.0f, 0.0f, 0.0f, 0.0f);
    float weights = 0.0f;
    const float dstep = SFO_TAU/(float)directions, wstep = 1.0f/(float)steps;
    for (float direction = 0.0f; direction < SFO_TAU; direction += dstep) {
        for (float walk = wstep; walk < 1.0f; walk += wstep) {
            float ox = ((sfo_cos(direction)*radius)*walk)/2000.0f;
            float oy = ((sfo_sin(direction)*radius)*walk)/2000.0f;
            v4 smp = sample(image, V2(stuv.x + ox, stuv.y + oy));
            float dx = ox - 0.0f, dy = oy - 0.0f;                          /* distance(offset, vec2(0)) */
            float weight = 1.0f - sfo_sqrt(dx*dx + dy*dy)/radius;
            color = v4_axpy(color, smp, weight);
            weights += weight;
        }
    }
    return V4(color.x/weights, color.y/weights, color.z/weights, color.w/weights);
}

/* examples/basic/shaders/multipass.frag:28-45 */
static v4 frag_multipass(const frag_in* f) {
    v4 out = V4(0.0f, 0.0f, 0.0f, 0.0f);
    switch (f->u->iLayer) {
        case 0:
            out = stexture(&f->tex[SFO_TEX_BACKGROUND], f->stuv);
            break;
        case 1: {
            const sfo_texture* iScreen0x0 = &f->tex[SFO_TEX_HISTORY];
            out = sample(iScreen0x0, f->astuv);
            if (f->gluv.x < 0.0f) out.x = 1.0f - out.x;
            else out = multipass_blur(iScreen0x0, f->astuv, 5.0f, 8, 8);
            break;
        }
        default: break;
    }
    out.w = 1.0f;
    return out;
}

/* examples/basic/shaders/motionblur.frag:1-18; user[0] = iScreenTemporal */
static v4 frag_motionblur(const frag_in* f) {
    v4 out = V4(0.0f, 0.0f, 0.0f, 0.0f);
    if (f->u->iLayer == 0) {
        camera_t cam = get_camera(f);
        out = stexture(&f->tex[SFO_TEX_BACKGROUND], cam.stuv);
    } else if (f->u->iLayer == 1) {
        int T = (int)f->u->user[0];
        v4 color = V4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int i = 0; i < T; i++) {
            float factor = sfo_smoothstep(1.0f, 0.0f, (float)i/(float)T);
            v4 past = (i < SFO_TEX_HISTORY_DEPTH) ? sample(&f->tex[SFO_TEX_HISTORY + i], f->astuv) : V4(0.0f, 0.0f, 0.0f, 0.0f);
            color = v4_axpy(color, past, factor);
        }
        out = V4((2.0f*color.x)/(float)T, (2.0f*color.y)/(float)T, (2.0f*color.z)/(float)T, (2.0f*color.w)/(float)T);
    }
    out.w = 1.0f;
    return out;
}

/* examples/basic/shaders/life/simulation.glsl:7-54; user[0..1] = iLifeSize, user[2] = iLifePeriod */
static v4 frag_life_simulation(const frag_in* f) {
    static const int alive[9] = {0, 0, 1, 1, 0, 0, 0, 0, 0};
    static const int dead[9] = {0, 0, 0, 1, 0, 0, 0, 0, 0};
    const sfo_texture* iLife1x0 = &f->tex[SFO_TEX_HISTORY + 1];
    v4 out = V4(0.0f, 0.0f, 0.0f, 0.0f);
    int period = (int)f->u->user[2];
    if ((f->u->iFrame % period) != 0) {
        out.x = sample(iLife1x0, f->astuv).x;
        out.w = 1.0f;
        return out;
    }
    int pixel_x = sfo_to_int(f->astuv.x*f->u->user[0]);
    int pixel_y = sfo_to_int(f->astuv.y*f->u->user[1]);
    int near = 0, current = 0;
    for (int x = -1; x <= 1; x++)
        for (int y = -1; y <= 1; y++) {
            int cell = texel_fetch(iLife1x0, pixel_x + x, pixel_y + y).x > 0.5f ? 1 : 0;
            if (x == 0 && y == 0) current = cell;
            else near += cell;
        }
    out.x = (float)((current == 1) ? alive[near] : dead[near]);
    out.w = 1.0f;
    return out;
}

/* shaderflow.glsl:210-218 */
static v3 palette4(float t, v3 A, v3 B, v3 C, v3 D) {
    v3 lo, hi; float k;
    if (t < 0.25f) { lo = A; hi = B; k = t*4.0f; }
    else if (t < 0.5f) { lo = B; hi = C; k = (t - 0.25f)*4.0f; }
    else { lo = C; hi = D; k = (t - 0.5f)*4.0f; }
    return V3(sfo_mix(lo.x, hi.x, k), sfo_mix(lo.y, hi.y, k), sfo_mix(lo.z, hi.z, k));
}
static const v3 MAGMA[4] = {{0.01060815f, 0.01808215f, 0.10018654f}, {0.38092887f, 0.12061482f, 0.32506528f},
                            {0.79650140f, 0.10506637f, 0.31063031f}, {0.95922872f, 0.53307513f, 0.37488950f}};

/* examples/basic/shaders/life/visuals.glsl:6-41 */
static v4 frag_life_visuals(const frag_in* f) {
    camera_t cam = get_camera(f);
    if (cam.out_of_bounds) return V4(MAGMA[0].x, MAGMA[0].y, MAGMA[0].z, 1.0f);
    float exponent = 1.3f;
    float area = 1.0f/(exponent + 1.0f);
    static const float base[5] = {1.0f, 0.8f, 0.6f, 0.4f, 0.2f};
    float life = 0.0f;
    for (int t = 0; t < 5; t++) {
        float r = stexture(&f->tex[SFO_TEX_HISTORY + t], cam.stuv).x;
        life += (t == 0) ? r : r*sfo_pow(base[t], exponent);
    }
    life /= (5.0f*area);
    v3 c = palette4(life, MAGMA[0], MAGMA[1], MAGMA[2], MAGMA[3]);
    return V4(c.x, c.y, c.z, 1.0f);
}

/* examples/basic/shaders/video.frag:1-6; iVideo = iVideo0x0 (texture.py:355-356) */
static v4 frag_video(const frag_in* f) {
    camera_t cam = get_camera(f);
    v4 c = stexture(&f->tex[SFO_TEX_HISTORY], cam.stuv);
    c.w = 1.0f;
    return c;
}

/* ---------------------------------------------------------------------------------------------- */
/* Remaining example fragments */

/* shaderflow.glsl:290-293 */
static float sdBox(v3 origin, v3 point, v3 size) {
    v3 d = V3(sfo_abs(origin.x - point.x) - size.x/2.0f, sfo_abs(origin.y - point.y) - size.y/2.0f, sfo_abs(origin.z - point.z) - size.z/2.0f);
    v3 q = V3(sfo_max(d.x, 0.0f), sfo_max(d.y, 0.0f), sfo_max(d.z, 0.0f));
    return sfo_min(sfo_max(d.x, sfo_max(d.y, d.z)), 0.0f) + sfo_sqrt(v3_dot(q, q));
}

/* examples/basic/shaders/raymarch.frag:5-59 */
static v4 frag_raymarch(const frag_in* f) {
    const int MAX_STEPS = 100; const float MAX_DIST = 100.0f, MIN_DIST = 0.001f;
    camera_t cam = get_camera(f);
    v3 dir = v3_sub(cam.target, cam.origin);
    float norm = sfo_sqrt(v3_dot(dir, dir));
    v3 forward = V3(dir.x/norm, dir.y/norm, dir.z/norm);
    float traveled = 0.0f, walk = 0.0f;
    int steps;
    for (steps = 0; steps < MAX_STEPS; steps++) {
        v3 point = v3_add(cam.origin, v3_scale(forward, traveled));
        float sdf = 2.0f*MAX_DIST;
        for (int i = 2; i < 8; i++) sdf = sfo_min(sdf, sdBox(point, V3(0.0f, 0.0f, (float)i), V3((float)(i - 1), (float)(i - 1), (float)(i - 1))));
        walk = sdf;
        traveled += walk;
        if (walk < MIN_DIST || walk > MAX_DIST) break;
    }
    float g = 1.0f - sfo_sqrt((float)steps)*0.1f;
    return V4(g, g, g, 1.0f);
}

/* examples/fractals/shaders/mandelbrot.frag:1-30 */
static v4 frag_mandelbrot(const frag_in* f) {
    camera_t cam = get_camera(f);
    float t = 0.0f;
    if (!cam.out_of_bounds) {
        float zx = cam.gluv.x - 0.5f, zy = cam.gluv.y - 0.0f;
        float cx = zx, cy = zy;
        int quality = sfo_to_int(1000.0f*f->u->iQuality);
        int iter = 0;
        for (; iter < quality; iter++) {
            if (sfo_sqrt(zx*zx + zy*zy) > 3.0f) break;
            float nx = (zx*zx - zy*zy) + cx;
            float ny = (zx*zy + zy*zx) + cy;
            zx = nx; zy = ny;
        }
        t = sfo_pow(1.0f - (float)iter/(float)quality, 20.0f);
    }
    v3 c = palette4(t, MAGMA[0], MAGMA[1], MAGMA[2], MAGMA[3]);
    return V4(c.x, c.y, c.z, 1.0f);
}

/* examples/fractals/shaders/tetration.frag:1-54 */
typedef struct { float x, y, r, t; } complex_t;
static v4 frag_tetration(const frag_in* f) {
    camera_t cam = get_camera(f);
    complex_t C = {cam.gluv.x, cam.gluv.y, 0.0f, 0.0f};
    C.r = sfo_sqrt(C.x*C.x + C.y*C.y);
    C.t = sfo_atan2(C.y, C.x);
    complex_t Z = C;
    const int MAX_STEPS = 67;
    int it;
    for (it = 0; it < MAX_STEPS; it++) {
        complex_t W;                                                      /* ComplexNumberPower(C, Z) :20-25 */
        W.r = sfo_pow(C.r, Z.x)*sfo_exp(-Z.y*C.t);
        W.t = Z.y*sfo_log(C.r) + (Z.x*C.t);
        W.x = W.r*sfo_cos(W.t);
        W.y = W.r*sfo_sin(W.t);
        Z = W;
        if (Z.r > 100.0f) break;
    }
    float k = (float)(it/MAX_STEPS);
    float theta = atan2_0_tau(Z.y, Z.x)/SFO_TAU;
    v3 c = hsv2rgb(theta, 1.0f, k);
    return V4(c.x, c.y, c.z, 1.0f);
}

static v4 shade(int fragment, const frag_in* f) {
    switch (fragment) {
        case SFO_FRAG_DEFAULT: return frag_default(f);
        case SFO_FRAG_VISUALIZER: return frag_visualizer(f);
        case SFO_FRAG_BARS: return frag_bars(f);
        case SFO_FRAG_WAVEFORM: return frag_waveform(f);
        case SFO_FRAG_MULTI_CHILD: return frag_multi_child(f);
        case SFO_FRAG_MULTI_MAIN: return frag_multi_main(f);
        case SFO_FRAG_SHADERTOY: return frag_shadertoy(f);
        case SFO_FRAG_DYNAMICS: return frag_dynamics(f);
        case SFO_FRAG_AUDIO: return frag_audio(f);
        case SFO_FRAG_MULTIPASS: return frag_multipass(f);
        case SFO_FRAG_MOTIONBLUR: return frag_motionblur(f);
        case SFO_FRAG_LIFE_SIMULATION: return frag_life_simulation(f);
        case SFO_FRAG_LIFE_VISUALS: return frag_life_visuals(f);
        case SFO_FRAG_VIDEO: return frag_video(f);
        case SFO_FRAG_RAYMARCH: return frag_raymarch(f);
        case SFO_FRAG_MANDELBROT: return frag_mandelbrot(f);
        case SFO_FRAG_TETRATION: return frag_tetration(f);
        default: return frag_missing(f);
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* Row-band drivers */

typedef struct {
    int kind;                 /* 0 render, 1 resolve */
    int fragment; const sfo_uniforms* u; const sfo_texture* tex;
    int wr, hr, w, h, subsample, y0, y1;
    const uint8_t* screen; uint8_t* out;
    int out_components, out_dtype;   /* render target format: RGBA8 unless sfo_render_to says otherwise */
} job_t;

static void render_rows(const job_t* jb) {
    frag_in f; f.u = jb->u; f.tex = jb->tex;
    int mipmapped = 0;
    for (int k = 0; k < SFO_TEX_SLOTS; k++) mipmapped |= (jb->tex[k].data && jb->tex[k].filter >= SFO_LINEAR_MIPMAP && jb->tex[k].levels > 1);
    for (int j = jb->y0; j < jb->y1; j++) {
        for (int i = 0; i < jb->wr; i++) {
            if (mipmapped) {                                            /* the quad's other column and other row first (see sample_mipmapped) */
                g_lod.mode = LOD_RECORD_X; g_lod.count = 0;
                make_varyings(&f, i ^ 1, j, jb->wr, jb->hr);
                (void)shade(jb->fragment, &f);
                g_lod.mode = LOD_RECORD_Y; g_lod.count = 0;
                make_varyings(&f, i, j ^ 1, jb->wr, jb->hr);
                (void)shade(jb->fragment, &f);
                g_lod.mode = LOD_REPLAY; g_lod.count = 0;
            }
            make_varyings(&f, i, j, jb->wr, jb->hr);
            v4 c = shade(jb->fragment, &f);
            g_lod.mode = LOD_OFF;
            const float channel[4] = {c.x, c.y, c.z, c.w};
            const int n = jb->out_components;
            for (int k = 0; k < n; k++) {
                if (jb->out_dtype == SFO_F32) ((float*)jb->out)[((int64_t)j*jb->wr + i)*n + k] = channel[k];
                else if (jb->out_dtype == SFO_F16) ((uint16_t*)jb->out)[((int64_t)j*jb->wr + i)*n + k] = float_to_half(channel[k]);
                else jb->out[((int64_t)j*jb->wr + i)*n + k] = to_unorm8(channel[k]);
            }
        }
    }
}

/* fragment/final.glsl:1-33; iScreen is RGBA8, linear, repeat(False) (scene.py:192-194, texture.py:108-112) */
static void resolve_rows(const job_t* jb) {
    sfo_texture screen = { jb->screen, jb->wr, jb->hr, 4, SFO_U8, SFO_LINEAR, 0, 0, 0, NULL };
    const int kernel = jb->subsample;
    const float resx = (float)jb->w, resy = (float)jb->h;               /* iResolution := scene.resolution, shader.py:394 */
    for (int j = jb->y0; j < jb->y1; j++) {
        for (int i = 0; i < jb->w; i++) {
            v2 astuv0 = V2(((float)i + 0.5f)/(float)jb->w, ((float)j + 0.5f)/(float)jb->h);
            v2 agluv = V2(astuv0.x*2.0f - 1.0f, astuv0.y*2.0f - 1.0f);
            v2 astuv = gluv2stuv(agluv);
            v3 rgb;
            if (kernel == 1) {                                          /* :6-10 */
                v4 c = sample(&screen, astuv);
                rgb = V3(c.x, c.y, c.z);
            } else {
                v3 acc = V3(0.0f, 0.0f, 0.0f);                          /* :13 */
                v2 pixel_size = V2(1.0f/resx, 1.0f/resy);               /* :17 */
                v2 corner = V2(astuv.x - (pixel_size.x/2.0f), astuv.y - (pixel_size.y/2.0f));        /* :20 */
                v2 origin = V2(corner.x + (pixel_size.x/(float)kernel)/2.0f, corner.y + (pixel_size.y/(float)kernel)/2.0f);   /* :21 */
                for (int x = 0; x < kernel; x++) {
                    for (int y = 0; y < kernel; y++) {
                        v2 offset = V2((pixel_size.x/(float)kernel)*(float)x, (pixel_size.y/(float)kernel)*(float)y);   /* :25 */
                        v4 c = sample(&screen, V2(origin.x + offset.x, origin.y + offset.y));
                        acc.x += c.x; acc.y += c.y; acc.z += c.z;       /* :26 */
                    }
                }
                float n = (float)(kernel*kernel);
                rgb = V3(acc.x/n, acc.y/n, acc.z/n);                    /* :31 */
            }
            uint8_t* px = jb->out + ((int64_t)j*jb->w + i)*3;
            px[0] = to_unorm8(rgb.x); px[1] = to_unorm8(rgb.y); px[2] = to_unorm8(rgb.z);
        }
    }
}

static void* job_main(void* arg) {
    const job_t* jb = (const job_t*)arg;
    if (jb->kind == 0) render_rows(jb); else resolve_rows(jb);
    return NULL;
}

static void run_jobs(job_t base, int threads) {
    int rows = base.y1 - base.y0;
    if (threads < 1) threads = 1;
    if (threads > rows) threads = rows > 0 ? rows : 1;
    if (threads == 1) { job_main(&base); return; }
    pthread_t* tid = (pthread_t*)malloc(sizeof(pthread_t)*threads);
    job_t* jobs = (job_t*)malloc(sizeof(job_t)*threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = base;
        jobs[t].y0 = base.y0 + (int)((int64_t)rows*t/threads);
        jobs[t].y1 = base.y0 + (int)((int64_t)rows*(t + 1)/threads);
        pthread_create(&tid[t], NULL, job_main, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    free(tid); free(jobs);
}

void sfo_render(int fragment, const sfo_uniforms* u, const sfo_texture* textures,
                int wr, int hr, int y0, int y1, int threads, uint8_t* out) {
    job_t jb = {0};
    jb.kind = 0; jb.fragment = fragment; jb.u = u; jb.tex = textures;
    jb.wr = wr; jb.hr = hr; jb.y0 = y0; jb.y1 = y1; jb.out = out;
    jb.out_components = 4; jb.out_dtype = SFO_U8;
    run_jobs(jb, threads);
}

void sfo_render_to(int fragment, const sfo_uniforms* u, const sfo_texture* textures,
                   int wr, int hr, int y0, int y1, int threads, int components, int dtype, void* out) {
    job_t jb = {0};
    jb.kind = 0; jb.fragment = fragment; jb.u = u; jb.tex = textures;
    jb.wr = wr; jb.hr = hr; jb.y0 = y0; jb.y1 = y1; jb.out = (uint8_t*)out;
    jb.out_components = components; jb.out_dtype = dtype;
    run_jobs(jb, threads);
}

void sfo_resolve(const uint8_t* screen, int wr, int hr, int w, int h, int subsample,
                 int y0, int y1, int threads, uint8_t* out) {
    job_t jb = {0};
    jb.kind = 1; jb.screen = screen; jb.wr = wr; jb.hr = hr; jb.w = w; jb.h = h;
    jb.subsample = subsample < 1 ? 1 : subsample; jb.y0 = y0; jb.y1 = y1; jb.out = out;
    run_jobs(jb, threads);
}

float sfo_test_math(int fn, float a, float b) {
    switch (fn) {
        case 0: return sfo_sin(a);
        case 1: return sfo_cos(a);
        case 2: return sfo_atan2(a, b);
        case 3: return sfo_atan(a);
        case 4: return sfo_log2(a);
        case 5: return sfo_exp2(a);
        case 6: return sfo_pow(a, b);
        case 7: return sfo_exp(a);
        case 8: return sfo_mod(a, b);
        case 9: return sfo_smoothstep(0.0f, a, b);
        case 10: return sfo_mix(0.25f, a, b);
        case 11: return sfo_sqrt(a);
        case 12: return sfo_log(a);
        default: return 0.0f;
    }
}
