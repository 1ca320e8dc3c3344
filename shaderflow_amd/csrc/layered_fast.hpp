// layered_fast.hpp — the second layer of the reference's layered / temporal example fragments at kernel rate, with the
// generic kernel's bits:
//   k_multipass_layer1    examples/basic/shaders/multipass.frag:34-42 — blur(iScreen0x0, astuv, 5, 8, 8) on the right half, 1 - red left
//   k_motionblur_layer1   examples/basic/shaders/motionblur.frag:8-15  — the weighted mean of iScreenTemporal history layers
//
// Round 4 ran both through k_render<PlainShader<…>>: 244 us (multipass) and 79 us (motionblur) per 1080p frame
// (profiles/r05_trace_layers.txt), an order of magnitude from any roof. What they spent it on does not depend on the pixel:
//   * multipass: the 9 x 7 tap offsets `(cos, sin)(direction) * radius * walk / 2000` and weights `1 - |offset| / radius` — two exact
//     (software) sine / cosine evaluations, a square root and a division per tap per pixel. Here a table of the frame (the host runs the
//     fragment's own float loops with the same sfmath.hpp functions: the same bits), and the texels come from an LDS tile that holds
//     them CONVERTED (unorm8 -> float once per texel instead of once per tap and neighbour: 16 conversions per tap less).
//   * motionblur: texel addressing and bilinear weights of `texture(iScreen{i}x0, astuv)` are the same for all layers (same size, same
//     sampler state: checked by the host), so they are computed once per pixel; the smoothstep factors once per launch.
// Every arithmetic operation that reaches the colour is the generic chain's, in its order (texture() of glsl.hpp: u*W - 0.5, floor, the
// four weights, fma chain; `color + sample*weight`; the final IEEE divisions), so tests/test_gpu_multipass.py keeps its array_equal.
#pragma once

#include "render_kernels.hpp"

namespace sf {

constexpr int LAYERED_MAX_TAPS = 96;                                   // 9 directions x 7 steps = 63 at the fragment's arguments

struct MultipassTaps {
    int count;
    float weights;                                                     // the fragment's running sum of the tap weights (its divisor)
    float reach_u, reach_v;                                            // max |offset| per axis (uv units): the tile's margin
    float ox[LAYERED_MAX_TAPS], oy[LAYERED_MAX_TAPS], weight[LAYERED_MAX_TAPS];
};

// blur() of multipass.frag:10-26 for uniform arguments: the offsets and weights in the order the fragment's float loops visit them
inline void multipass_tap_table(MultipassTaps& t, float radius, int directions, int steps) {
    t.count = 0; t.weights = 0.0f; t.reach_u = 0.0f; t.reach_v = 0.0f;
    for (float direction = 0.0f; direction < TAU; direction += TAU/(float)directions) {
        for (float walk = 1.0f/(float)steps; walk < 1.0f; walk += 1.0f/(float)steps) {
            const vec2 offset = vec2{sf::cos(direction), sf::sin(direction)}*radius*walk/2000.0f;
            const float weight = 1.0f - length(offset - vec2{0.0f, 0.0f})/radius;
            if (t.count < LAYERED_MAX_TAPS) { t.ox[t.count] = offset.x; t.oy[t.count] = offset.y; t.weight[t.count] = weight; }
            t.count++;
            t.weights += weight;
            t.reach_u = fmaxf(t.reach_u, fabsf(offset.x)); t.reach_v = fmaxf(t.reach_v, fabsf(offset.y));
        }
    }
}

#ifndef MP_ROWS_PER_THREAD
#define MP_ROWS_PER_THREAD 2
#endif
constexpr int MP_BLOCK_W = 64, MP_BLOCK_H = 8, MP_ROWS = MP_ROWS_PER_THREAD, MP_THREADS = MP_BLOCK_W*MP_BLOCK_H/MP_ROWS;

// 64 x 8 pixels per block, MP_ROWS of a column per thread. The tile: the texels of iScreen0x0 under the block plus the blur's reach, wrapped as the
// sampler wraps (so a tap indexes it without any clamp), four floats per texel (texel() of glsl.hpp applied once), in dynamic LDS.
__global__ __launch_bounds__(MP_THREADS) void k_multipass_layer1(const RenderArgs a, const MultipassTaps* __restrict__ taps, int tile_w, int tile_h) {
    extern __shared__ __attribute__((aligned(16))) float4 mp_tile[];
    __shared__ float tap_x[LAYERED_MAX_TAPS], tap_y[LAYERED_MAX_TAPS], tap_w[LAYERED_MAX_TAPS];
    const int tid = threadIdx.y*MP_BLOCK_W + threadIdx.x;
    const int i = blockIdx.x*MP_BLOCK_W + threadIdx.x, j = blockIdx.y*MP_BLOCK_H + threadIdx.y;
    const Tex first = a.tex[TEX_HISTORY];                                                  // iScreen0x0
    const int count = taps->count;
    const float W = (float)first.width, H = (float)first.height;
    Frag f; f.u = &a.u; f.tex = a.tex; f.history = a.tex + TEX_HISTORY;

    // the block's window: its two corner pixels bound every tap (texel coordinates are monotone in the pixel index)
    const int i_lo = blockIdx.x*MP_BLOCK_W, i_hi = min(i_lo + MP_BLOCK_W, a.wr) - 1, j_lo = blockIdx.y*MP_BLOCK_H, j_hi = min(j_lo + MP_BLOCK_H, a.hr) - 1;
    Frag corner = f;
    make_varyings(corner, i_lo, j_lo, a.wr, a.hr, a.aspect, a.inv_wr, a.inv_hr);
    const vec2 uv_lo = corner.astuv;
    const bool any_right = [&] { Frag c = f; make_varyings(c, i_hi, j_lo, a.wr, a.hr, a.aspect, a.inv_wr, a.inv_hr); return !(c.gluv.x < 0.0f); }();
    make_varyings(corner, i_hi, j_hi, a.wr, a.hr, a.aspect, a.inv_wr, a.inv_hr);
    const vec2 uv_hi = corner.astuv;
    const int x0 = (int)::floorf((uv_lo.x - taps->reach_u)*W - 0.5f) - 1, y0 = (int)::floorf((uv_lo.y - taps->reach_v)*H - 0.5f) - 1;
    const int x1 = (int)::floorf((uv_hi.x + taps->reach_u)*W - 0.5f) + 2, y1 = (int)::floorf((uv_hi.y + taps->reach_v)*H - 0.5f) + 2;
    const int tw = x1 - x0 + 1, th = y1 - y0 + 1;
    const bool tiled = any_right && tw <= tile_w && th <= tile_h;                          // (a block of the left half reads one texel per pixel: no tile)
    if (tiled) {
        for (int e = tid; e < tw*th; e += MP_THREADS) {
            const int ty = e / tw, tx = e - ty*tw;
            const vec4 c = texel(first, wrap_texel(x0 + tx, first.width, first.repeat_x), wrap_texel(y0 + ty, first.height, first.repeat_y));
            mp_tile[e] = make_float4(c.x, c.y, c.z, c.w);
        }
    }
    if (any_right) for (int e = tid; e < count && e < LAYERED_MAX_TAPS; e += MP_THREADS) { tap_x[e] = taps->ox[e]; tap_y[e] = taps->oy[e]; tap_w[e] = taps->weight[e]; }
    __syncthreads();
    if (i >= a.wr) return;

    // the tile's byte offset of texel (fu, fv) in FLOAT arithmetic (small integers: exact), one conversion instead of two plus an integer
    // multiply: 16*((fv - y0)*tw + (fu - x0)) = fv*tw16 + (fu*16 - origin)
    const float tw16 = (float)(tw*16), origin = (float)((y0*tw + x0)*16);
    const char* tile_bytes = (const char*)mp_tile;
    const int up = tw*16;
    // MP_ROWS pixels per thread (rows threadIdx.y and threadIdx.y + MP_BLOCK_H/MP_ROWS of the block): two independent chains of LDS
    // reads and multiply-adds per tap, one read of the tap table for both
    vec2 astuv[MP_ROWS]; vec4 color[MP_ROWS]; bool inside[MP_ROWS], left[MP_ROWS]; bool any_blur = false;
#pragma unroll
    for (int q = 0; q < MP_ROWS; q++) {
        const int jq = j + q*(MP_BLOCK_H/MP_ROWS);
        inside[q] = jq < a.hr;
        make_varyings(f, i, inside[q] ? jq : j, a.wr, a.hr, a.aspect, a.inv_wr, a.inv_hr);
        astuv[q] = f.astuv; left[q] = f.gluv.x < 0.0f;
        color[q] = {0.0f, 0.0f, 0.0f, 0.0f};                                               // :11
        any_blur = any_blur || (inside[q] && !left[q]);
    }
    if (any_blur) {
        for (int k = 0; k < count; k++) {                                                  // (not unrolled: the optimizer declines, and said so on every build)
            const vec2 offset = {tap_x[k], tap_y[k]};
            const float weight = tap_w[k];
#pragma unroll
            for (int q = 0; q < MP_ROWS; q++) {
                const vec2 uv = astuv[q] + offset;                                         // :18 texture(image, stuv + offset)
                vec4 sample;
                if (tiled) {
                    // texture() of glsl.hpp, LINEAR: the same operations; the four texels from the tile (already wrapped, already floats)
                    const float u = uv.x*W, v = uv.y*H;
                    const float ub = u - 0.5f, vb = v - 0.5f;
                    const float fu = ::floorf(ub), fv = ::floorf(vb);
                    const float ax = ub - fu, by = vb - fv;
                    const char* cell = tile_bytes + (int)fmaf(fv, tw16, fmaf(fu, 16.0f, -origin));
                    const float4 t00 = *(const float4*)cell, t10 = *(const float4*)(cell + 16), t01 = *(const float4*)(cell + up), t11 = *(const float4*)(cell + up + 16);
                    const float na = 1.0f - ax, nb = 1.0f - by;
                    const float w00 = na*nb, w10 = ax*nb, w01 = na*by, w11 = ax*by;
                    sample = {bilerp(w00, w10, w01, w11, t00.x, t10.x, t01.x, t11.x), bilerp(w00, w10, w01, w11, t00.y, t10.y, t01.y, t11.y),
                              bilerp(w00, w10, w01, w11, t00.z, t10.z, t01.z, t11.z), 1.0f};   // (alpha is overwritten below: never computed)
                } else {
                    sample = texture(first, uv);
                }
                color[q] = color[q] + sample*weight;                                       // :20
            }
        }
    }
#pragma unroll
    for (int q = 0; q < MP_ROWS; q++) {
        if (!inside[q]) continue;
        vec4 col;
        if (left[q]) {                                                                     // multipass.frag:35, 38-39
            col = texture(first, astuv[q]);
            col.x = 1.0f - col.x;
        } else {
            col = color[q]/taps->weights;                                                  // :25
        }
        col.w = 1.0f;                                                                      // :44
        store_target(a, blockIdx.z, i, j + q*(MP_BLOCK_H/MP_ROWS), col);
    }
}

// ---- motionblur.frag:8-15 -----------------------------------------------------------------------------------------------------------
// The host has checked that the `temporal` history layers have one size and one sampler state (RGBA8, LINEAR): the texel indices and the
// four bilinear weights of texture(iScreen{i}x0, astuv) are then one computation per pixel. factor[i] = smoothstep(1, 0, i/temporal) by
// the host with the fragment's own operations.
struct MotionblurArgs {
    const uint32_t* layer[TEX_HISTORY_DEPTH];                          // RGBA8 texels of history[i]
    float factor[TEX_HISTORY_DEPTH];
    int temporal;
};

__global__ __launch_bounds__(256) void k_motionblur_layer1(const RenderArgs a, const MotionblurArgs m) {
    const int i = blockIdx.x*64 + threadIdx.x, j = blockIdx.y*4 + threadIdx.y;
    if (i >= a.wr || j >= a.hr) return;
    Frag f; f.u = &a.u; f.tex = a.tex; f.history = a.tex + TEX_HISTORY;
    make_varyings(f, i, j, a.wr, a.hr, a.aspect, a.inv_wr, a.inv_hr);
    const Tex& t = a.tex[TEX_HISTORY];
    // texture() of glsl.hpp for a LINEAR sampler, once for all layers
    const float u = f.astuv.x*(float)t.width, v = f.astuv.y*(float)t.height;
    const float ub = u - 0.5f, vb = v - 0.5f;
    const float fu = ::floorf(ub), fv = ::floorf(vb);
    const float ax = ub - fu, by = vb - fv;
    const int i0 = wrap_texel((int)fu, t.width, t.repeat_x), i1 = wrap_texel((int)fu + 1, t.width, t.repeat_x);
    const int j0 = wrap_texel((int)fv, t.height, t.repeat_y), j1 = wrap_texel((int)fv + 1, t.height, t.repeat_y);
    const uint32_t e00 = (uint32_t)j0*(uint32_t)t.width + (uint32_t)i0, e10 = (uint32_t)j0*(uint32_t)t.width + (uint32_t)i1;
    const uint32_t e01 = (uint32_t)j1*(uint32_t)t.width + (uint32_t)i0, e11 = (uint32_t)j1*(uint32_t)t.width + (uint32_t)i1;
    const float na = 1.0f - ax, nb = 1.0f - by;
    const float w00 = na*nb, w10 = ax*nb, w01 = na*by, w11 = ax*by;
    vec3 color = {0.0f, 0.0f, 0.0f};                                                       // (alpha is overwritten: fragColor.a = 1)
    auto channel = [](uint32_t word, int shift) { return unorm8_to_float((float)((word >> shift) & 255u)); };
#pragma unroll 2
    for (int k = 0; k < m.temporal; k++) {
        const uint32_t* p = m.layer[k];
        const uint32_t q00 = p[e00], q10 = p[e10], q01 = p[e01], q11 = p[e11];
        const vec3 sample = {bilerp(w00, w10, w01, w11, channel(q00, 0), channel(q10, 0), channel(q01, 0), channel(q11, 0)),
                             bilerp(w00, w10, w01, w11, channel(q00, 8), channel(q10, 8), channel(q01, 8), channel(q11, 8)),
                             bilerp(w00, w10, w01, w11, channel(q00, 16), channel(q10, 16), channel(q01, 16), channel(q11, 16))};
        color = color + sample*m.factor[k];                                                // :12
    }
    const vec3 out = (color*2.0f)/(float)m.temporal;                                       // :14
    store_target(a, blockIdx.z, i, j, vec4{out.x, out.y, out.z, 1.0f});
}

}  // namespace sf
