// resolve_fast.hpp — final.glsl as a pass (render_kernels.hpp k_resolve) with its coordinate arithmetic taken out of the pixels.
//
// final.glsl:6-31 taps iScreen kernel x kernel times per output pixel; the tap coordinates are separable — the x coordinate of
// tap (x, y) of pixel (i, j) depends on (i, x) only, the y coordinate on (j, y) only — and so is everything texture() derives
// from them before it touches a texel: the two texel indices and the two weights per axis. k_resolve_axis evaluates k_resolve's /
// final_glsl's / texture()'s operations for ONE coordinate per (pixel index, tap) — the same sequence, so the same bits — into a
// table of w*kernel + h*kernel entries; k_resolve_fast then only multiplies the weights, blends and accumulates in final.glsl's
// order (x outer, y inner). The iScreen window of a block of 64 x 4 pixels is staged once in LDS as floats (c/255 per channel,
// glsl.hpp unorm8_to_float — once per texel instead of once per tap), the finished RGB8 rows leave through LDS as 16-byte
// stores (k_resolve wrote three single bytes per pixel): C2's resolve pass 1.84 ms -> see DESIGN §7 per 60 frames of 1920x1080.
// Bit-identical to k_resolve by construction (tests/test_gpu_pixels.py asserts array_equal against the oracle's final.glsl).
#pragma once

#include "render_kernels.hpp"

namespace sf {

struct ResolveTables {
    const int4* columns;             // [w][kernel] = { i0, i1, bits(a), bits(1 - a) }
    const int4* rows;                // [h][kernel] = { j0, j1, bits(b), bits(1 - b) }
};
constexpr int RESOLVE_WINDOW_TEXELS = 2304;      // at most 36 KB of float4 per block; the launch sizes the LDS window to the block's footprint

template <int AXIS>
__global__ __launch_bounds__(128) void k_resolve_axis(const ResolveArgs a, int4* table) {
    const int n = AXIS == 0 ? a.w : a.h;
    const int index = blockIdx.x*128 + threadIdx.x;
    if (index >= n) return;
    const int size = AXIS == 0 ? a.screen.width : a.screen.height;
    const int repeat = AXIS == 0 ? a.screen.repeat_x : a.screen.repeat_y;
    const float centre = ((float)index + 0.5f)/(float)n;                                      // k_resolve
    const float astuv = ((centre*2.0f - 1.0f) + 1.0f)/2.0f;                                   // gluv2stuv(centre*2 - 1)
    const int kernel = a.subsample;
    const float pixel_size = 1.0f/(float)n;                                                   // final.glsl:17
    const float corner = astuv - (pixel_size/2.0f);                                           // :20
    const float origin = corner + (pixel_size/(float)kernel)/2.0f;                            // :21
    for (int tap = 0; tap < kernel; tap++) {
        const float offset = (pixel_size/(float)kernel)*(float)tap;                           // :25
        const float coordinate = (kernel == 1) ? astuv : origin + offset;                     // :6-10 / :26
        const float u = coordinate*(float)size;                                               // glsl.hpp texture()
        const float ub = u - 0.5f;
        const float fu = ::floorf(ub);
        const float weight = ub - fu;
        const int i0 = wrap_texel((int)fu, size, repeat), i1 = wrap_texel((int)fu + 1, size, repeat);
        table[(long)index*kernel + tap] = make_int4(i0, i1, __float_as_int(weight), __float_as_int(1.0f - weight));
    }
}

template <int KERNEL>
__global__ __launch_bounds__(256) void k_resolve_fast(const ResolveArgs a, const ResolveTables t, const int window_texels) {
    constexpr int BW = 64, BH = 4;
    extern __shared__ __attribute__((aligned(16))) float4 window[];                              // window_texels entries
    __shared__ __attribute__((aligned(16))) uint8_t staged[BH][BW*3];
    const int tid = threadIdx.y*BW + threadIdx.x;
    const int i_first = blockIdx.x*BW, j_first = blockIdx.y*BH;
    const int i_last = min(i_first + BW, a.w) - 1, j_last = min(j_first + BH, a.h) - 1;
    const int i = min(i_first + (int)threadIdx.x, a.w - 1), j = min(j_first + (int)threadIdx.y, a.h - 1);
    const uint32_t* screen = (const uint32_t*)((const char*)a.screen.data + (long)blockIdx.z*a.screen_frame_stride);
    // the block's window of iScreen: indices do not decrease with the pixel or the tap (the host takes this path for clamped
    // textures only), so the first pixel's first tap and the last pixel's last tap bound it
    const int x0 = t.columns[(long)i_first*KERNEL].x, x1 = t.columns[(long)i_last*KERNEL + KERNEL - 1].y;
    const int y0 = t.rows[(long)j_first*KERNEL].x, y1 = t.rows[(long)j_last*KERNEL + KERNEL - 1].y;
    const int tw = x1 - x0 + 1, th = y1 - y0 + 1;
    const bool tiled = tw*th <= window_texels;
    if (tiled) {
        for (int k = tid; k < tw*th; k += BW*BH) {
            const int ty = k / tw, tx = k - ty*tw;
            const uint32_t w = screen[(long)(y0 + ty)*a.screen.width + (x0 + tx)];
            window[k] = make_float4(unorm8_to_float((float)(w & 255u)), unorm8_to_float((float)((w >> 8) & 255u)), unorm8_to_float((float)((w >> 16) & 255u)), 0.0f);
        }
    }
    __syncthreads();
    float r = 0.0f, g = 0.0f, b = 0.0f;
    auto taps = [&](auto fetch) {
#pragma unroll
        for (int x = 0; x < KERNEL; x++) {
            const int4 c = t.columns[(long)i*KERNEL + x];
            const float wa = __int_as_float(c.z), na = __int_as_float(c.w);
#pragma unroll
            for (int y = 0; y < KERNEL; y++) {
                const int4 q = t.rows[(long)j*KERNEL + y];
                const float wb = __int_as_float(q.z), nb = __int_as_float(q.w);
                const float w00 = na*nb, w10 = wa*nb, w01 = na*wb, w11 = wa*wb;                  // glsl.hpp texture()
                const float4 t00 = fetch(c.x, q.x), t10 = fetch(c.y, q.x), t01 = fetch(c.x, q.y), t11 = fetch(c.y, q.y);
                const float tr = bilerp(w00, w10, w01, w11, t00.x, t10.x, t01.x, t11.x);
                const float tg = bilerp(w00, w10, w01, w11, t00.y, t10.y, t01.y, t11.y);
                const float tb = bilerp(w00, w10, w01, w11, t00.z, t10.z, t01.z, t11.z);
                if (KERNEL == 1) { r = tr; g = tg; b = tb; }                                     // final.glsl:6-10
                else { r = r + tr; g = g + tg; b = b + tb; }                                     // :26
            }
        }
    };
    if (tiled) {
        const float4* base = window - (y0*tw + x0);
        taps([&](int x, int y) { return base[y*tw + x]; });
    } else {
        taps([&](int x, int y) {
            const uint32_t w = screen[(long)y*a.screen.width + x];
            return make_float4(unorm8_to_float((float)(w & 255u)), unorm8_to_float((float)((w >> 8) & 255u)), unorm8_to_float((float)((w >> 16) & 255u)), 0.0f);
        });
    }
    if (KERNEL > 1) { const float count = (float)(KERNEL*KERNEL); r = r/count; g = g/count; b = b/count; }   // :31
    uint8_t* s = &staged[threadIdx.y][threadIdx.x*3];
    s[0] = (uint8_t)unorm8(r); s[1] = (uint8_t)unorm8(g); s[2] = (uint8_t)unorm8(b);
    __syncthreads();
    uint8_t* out = a.out + (long)blockIdx.z*a.out_frame_stride;
    // full-width blocks of frames whose rows are whole 16-byte groups: one sweep of 16-byte stores for all the block's rows
    constexpr int GROUPS = BW*3/16;
    if ((BW*3) % 16 == 0 && i_first + BW <= a.w && (a.w*3) % 16 == 0 && ((uintptr_t)out & 15) == 0 && ((i_first*3) & 15) == 0) {
        for (int e = tid; e < BH*GROUPS; e += BW*BH) {
            const int row = e/GROUPS, c = e - row*GROUPS, jr = j_first + row;
            if (jr < a.h) stream_store16((uint4*)(out + (long)(a.top_down ? a.h - 1 - jr : jr)*a.w*3 + (long)i_first*3) + c, (const uint4*)&staged[0][0] + e);
        }
        return;
    }
#pragma unroll
    for (int row = 0; row < BH; row++) {
        const int jr = j_first + row;
        if (jr < a.h) store_rgb_row(out + (long)(a.top_down ? a.h - 1 - jr : jr)*a.w*3, i_first, a.w, staged[row], tid, BW*BH, BW);
    }
}

// ---- the two-pass configuration: iScreen the size of the output, kernel 2 (BASELINE config 2: 1920x1080 without SSAA) ---------------
// final.glsl's four taps then sit a quarter of a texel around the pixel's own texel centre: tap (x, y) blends texels
// (i - 1 + x, i + x) x (j - 1 + y, j + y) — a 3 x 3 neighbourhood, the middle texel in all four taps. k_resolve_fast reads sixteen
// float4 from LDS per pixel (its taps index the window through the tables, so nothing is reused) in blocks of 256 pixels that
// spend most of their life waiting: table entry -> window texels -> barrier -> per-tap table entries -> ...; 810 us per 60
// frames of 1080p = 1.1 TB/s of its 871 MB. Here a thread owns FOUR vertically adjacent pixels: the 3 x 6 texels under them are read
// once (4.5 LDS reads per pixel), the column's two tap entries sit in registers, the rows' entries are wave-uniform (a wave is one
// row group: scalar loads), a block is 64 x 16 pixels (19 KB of window: eight blocks per CU). The arithmetic per tap is
// k_resolve_fast's — the same weight products, the same bilerp, the same accumulation order (x outer, y inner) — so the bytes are
// the same: tests/test_gpu_pixels.py holds both to np.array_equal against the oracle's final.glsl.
// That the taps of vertically / horizontally adjacent pixels chain (tap 1's first texel is tap 0's second, the next pixel's tap 0
// starts where this one's tap 1 did) follows from the coordinates: tap 0 lands at texel index - 0.25, tap 1 at + 0.25, 1e-4 away
// from nothing; the kernel still takes every index and weight from the tables, it only trusts that equal indices are equal.
constexpr int TENT_BW = 64, TENT_BH = 16, TENT_ROWS_PER_THREAD = 4;
__global__ __launch_bounds__(256) void k_resolve_tent(const ResolveArgs a, const ResolveTables t) {
    constexpr int WINDOW = (TENT_BW + 2)*(TENT_BH + 2);
    __shared__ __attribute__((aligned(16))) float4 window[WINDOW];
    __shared__ __attribute__((aligned(16))) uint8_t staged[TENT_BH][TENT_BW*3];
    const int lane = threadIdx.x, group = __builtin_amdgcn_readfirstlane((int)threadIdx.y);      // 4 waves, wave g owns rows 4g .. 4g + 3 of the block
    const int tid = group*TENT_BW + lane;
    const int i_first = blockIdx.x*TENT_BW, j_first = blockIdx.y*TENT_BH;
    const int i_last = min(i_first + TENT_BW, a.w) - 1, j_last = min(j_first + TENT_BH, a.h) - 1;
    const int i = min(i_first + lane, a.w - 1);
    const uint32_t* screen = (const uint32_t*)((const char*)a.screen.data + (long)blockIdx.z*a.screen_frame_stride);
    const int x0 = t.columns[(long)i_first*2].x, x1 = t.columns[(long)i_last*2 + 1].y;
    const int y0 = t.rows[(long)j_first*2].x, y1 = t.rows[(long)j_last*2 + 1].y;
    const int tw = x1 - x0 + 1, th = y1 - y0 + 1;                    // <= 66 x 18 by construction (clamped at the frame's edges)
    for (int k = tid; k < tw*th; k += 256) {
        const int ty = k / tw, tx = k - ty*tw;
        const uint32_t w = screen[(long)(y0 + ty)*a.screen.width + (x0 + tx)];
        window[k] = make_float4(unorm8_to_float((float)(w & 255u)), unorm8_to_float((float)((w >> 8) & 255u)), unorm8_to_float((float)((w >> 16) & 255u)), 0.0f);
    }
    // the column's two taps: texel columns (ca.x, ca.y) and (cb.x, cb.y) with ca.y == cb.x
    const int4 ca = t.columns[(long)i*2], cb = t.columns[(long)i*2 + 1];
    const float wa0 = __int_as_float(ca.z), na0 = __int_as_float(ca.w), wa1 = __int_as_float(cb.z), na1 = __int_as_float(cb.w);
    const int cx[3] = {ca.x - x0, ca.y - x0, cb.y - x0};
    __syncthreads();
    const int j_group = j_first + group*TENT_ROWS_PER_THREAD;
    // texel rows under the four pixels
    float4 texel[TENT_ROWS_PER_THREAD + 2][3];
    int4 qa[TENT_ROWS_PER_THREAD], qb[TENT_ROWS_PER_THREAD];
#pragma unroll
    for (int m = 0; m < TENT_ROWS_PER_THREAD; m++) {
        const int j = min(j_group + m, a.h - 1);
        qa[m] = t.rows[(long)j*2]; qb[m] = t.rows[(long)j*2 + 1];       // wave-uniform addresses: scalar loads
    }
#pragma unroll
    for (int m = 0; m < TENT_ROWS_PER_THREAD + 2; m++) {
        // R[0] = the first row of pixel 0's tap 0, R[m + 1] = the second row of pixel m's tap 0 (= the first of its tap 1 = the first of
        // pixel m + 1's tap 0 while that pixel exists; past the frame's last row the entries repeat and nothing reads the result)
        const int row = (m == 0) ? qa[0].x : (m <= TENT_ROWS_PER_THREAD ? qa[m - 1].y : qb[TENT_ROWS_PER_THREAD - 1].y);
        const float4* line = window + (row - y0)*tw;
#pragma unroll
        for (int x = 0; x < 3; x++) texel[m][x] = line[cx[x]];
    }
#pragma unroll
    for (int m = 0; m < TENT_ROWS_PER_THREAD; m++) {
        float r = 0.0f, g = 0.0f, b = 0.0f;
#pragma unroll
        for (int x = 0; x < 2; x++) {
            const float wa = x ? wa1 : wa0, na = x ? na1 : na0;
#pragma unroll
            for (int y = 0; y < 2; y++) {
                const int4 q = y ? qb[m] : qa[m];
                const float wb = __int_as_float(q.z), nb = __int_as_float(q.w);
                const float w00 = na*nb, w10 = wa*nb, w01 = na*wb, w11 = wa*wb;                      // glsl.hpp texture()
                const float4 t00 = texel[m + y][x], t10 = texel[m + y][x + 1], t01 = texel[m + y + 1][x], t11 = texel[m + y + 1][x + 1];
                r = r + bilerp(w00, w10, w01, w11, t00.x, t10.x, t01.x, t11.x);                        // final.glsl:26
                g = g + bilerp(w00, w10, w01, w11, t00.y, t10.y, t01.y, t11.y);
                b = b + bilerp(w00, w10, w01, w11, t00.z, t10.z, t01.z, t11.z);
            }
        }
        r = r/4.0f; g = g/4.0f; b = b/4.0f;                                                            // :31
        uint8_t* s = &staged[group*TENT_ROWS_PER_THREAD + m][lane*3];
        s[0] = (uint8_t)unorm8(r); s[1] = (uint8_t)unorm8(g); s[2] = (uint8_t)unorm8(b);
    }
    __syncthreads();
    uint8_t* out = a.out + (long)blockIdx.z*a.out_frame_stride;
    constexpr int GROUPS = TENT_BW*3/16;
    if (i_first + TENT_BW <= a.w && (a.w*3) % 16 == 0 && ((uintptr_t)out & 15) == 0) {
        static_assert(TENT_BH*GROUPS <= 256, "one 16-byte group per thread");
        if (const int e = tid; e < TENT_BH*GROUPS) {
            const int row = e/GROUPS, c = e - row*GROUPS, jr = j_first + row;
            if (jr < a.h) stream_store16((uint4*)(out + (long)(a.top_down ? a.h - 1 - jr : jr)*a.w*3 + (long)i_first*3) + c, (const uint4*)&staged[0][0] + e);
        }
        return;
    }
#pragma unroll 1
    for (int row = 0; row < TENT_BH; row++) {
        const int jr = j_first + row;
        if (jr < a.h) store_rgb_row(out + (long)(a.top_down ? a.h - 1 - jr : jr)*a.w*3, i_first, a.w, staged[row], tid, 256, TENT_BW);
    }
}

}  // namespace sf
