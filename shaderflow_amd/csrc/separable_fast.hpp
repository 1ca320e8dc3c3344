// separable_fast.hpp — the light audio fragments (examples/basic/shaders/bars.frag:5-22, waveform.frag:5-19) at their roof.
//
// Both colour a sample by COMPARING one coordinate with values looked up through the other one:
//   bars.frag      texture(iSpectrogram, astuv.yx) with a one-column spectrogram depends on the sample COLUMN only; the
//                  comparisons and the blue ramp use astuv.y, a function of the sample ROW only;
//   waveform.frag  texture(iWaveform, (astuv.x, 0)) depends on the column only; abs(gluv.y) on the row only.
// So per frame the texture look-ups, square roots and divisions are done once per column (k_separable_axis) instead of once
// per supersample — 7 680 instead of 33.2 M at 4K 2xSSAA — with the generic chain's operations (same bits as fragments.hpp
// frag_bars / frag_waveform), and the fused kernel keeps what is two-dimensional: three comparisons, one multiply-add, the
// RGBA8 quantisation and final.glsl's resolve. One thread per OUTPUT pixel shades its 2 x 2 supersamples from two column and
// two row entries and walks SEP_ROWS rows with the column entries in registers; rows leave through LDS as 16-byte stores.
// What bounds it then is issue of the resolve arithmetic and the HBM write of the RGB8 frame (24.9 MB at 4K).
#pragma once

#include "render_kernels.hpp"

namespace sf {

enum : int { SEP_BARS = 0, SEP_WAVEFORM = 1 };
constexpr int SEP_ROWS = 4, SEP_PIXELS = 256;

struct SepTables {
    float4* columns;                     // [frame][wr]
    float4* rows;                        // [frame][hr]
};

// thread k < wr builds column k, thread wr + k builds row k, of frame blockIdx.y
template <int KIND>
__global__ __launch_bounds__(256) void k_separable_axis(const RenderArgs a, const SepTables t) {
    const int k = blockIdx.x*256 + threadIdx.x;
    if (k >= a.wr + a.hr) return;
    const int frame = blockIdx.y;
    Uniforms u; Tex tex[TEX_HISTORY];
    frame_view(a, frame, u, tex);
    const bool column = k < a.wr;
    const int index = column ? k : k - a.wr, n = column ? a.wr : a.hr;
    // vertex/default.glsl:1-17 for one coordinate (glsl.hpp make_varyings)
    const float centre = pixel_centre(index, n, column ? a.inv_wr : a.inv_hr);
    const float ag = centre*2.0f - 1.0f;
    const float g = ag*(column ? a.aspect : 1.0f);
    const float as = (ag + 1.0f)/2.0f;
    float4 e = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (KIND == SEP_BARS) {
        if (column) {
            // texture(iSpectrogram, astuv.yx): any astuv.y picks the one column, astuv.x the bin (bars.frag:9)
            const vec2 s = texture_xy(tex[TEX_SPECTROGRAM], vec2{0.5f, as});
            const vec2 intensity = {sf::sqrt(s.x)/120.0f, sf::sqrt(s.y)/120.0f};                                   // :10
            e = make_float4(intensity.x, intensity.y, (intensity.y + intensity.x)/2.0f, 0.4f*(intensity.x + intensity.y));   // :14-17
        } else {
            e = make_float4(as, 1.0f - as, 0.0f, 0.0f);
        }
    } else {
        if (column) {
            const vec2 w = texture_xy(tex[TEX_WAVEFORM], vec2{as, 0.0f});                                          // waveform.frag:6
            e = make_float4(w.x, w.y, (w.x + w.y)/2.0f, 0.0f);                                                     // :13-15
        } else {
            e = make_float4(sf::abs(g), 0.0f, 0.0f, 0.0f);                                                         // abs(gluv.y), :13
        }
    }
    (column ? t.columns + (long)frame*a.wr : t.rows + (long)frame*a.hr)[index] = e;
}

template <int KIND> __device__ __forceinline__ uint32_t separable_texel(const float4 c, const float4 r) {
    vec3 col;
    if (KIND == SEP_BARS) {                                           // bars.frag:11-18 on (intensity.x, .y, mean, 0.4*sum) and (astuv.y, 1 - astuv.y)
        col.x = (r.x < c.x) ? 1.0f : 0.0f;
        col.y = (r.x < c.y) ? 1.0f : 0.0f;
        col.z = ((r.x < c.z) ? 1.0f : 0.0f) + c.w*r.y;
    } else {                                                          // waveform.frag:10-15
        col.x = (r.x < c.x) ? 1.0f : 0.2f;
        col.y = (r.x < c.y) ? 1.0f : 0.2f;
        col.z = (r.x < c.z) ? 1.0f : 0.2f;
    }
    return pack_rgb8(col);
}

// S == 2. grid (ceil(w/SEP_PIXELS), ceil(h/SEP_ROWS), frames), block SEP_PIXELS threads.
template <int KIND>
__global__ __launch_bounds__(SEP_PIXELS) void k_separable_fused(const RenderArgs a, const SepTables t) {
    __shared__ __attribute__((aligned(16))) uint8_t staged[SEP_ROWS][SEP_PIXELS*3];
    const int frame = blockIdx.z;
    const int tid = threadIdx.x;
    const int px = blockIdx.x*SEP_PIXELS + tid;
    const bool inside = px < a.w;
    const float4* columns = t.columns + (long)frame*a.wr;
    const float4* rows = t.rows + (long)frame*a.hr;
    const int i0 = (2*px < a.wr) ? 2*px : a.wr - 1, i1 = (2*px + 1 < a.wr) ? 2*px + 1 : a.wr - 1;
    const float4 c0 = columns[i0], c1 = columns[i1];
#pragma unroll
    for (int r = 0; r < SEP_ROWS; r++) {
        const int py = blockIdx.y*SEP_ROWS + r;
        if (py >= a.h) break;
        const float4 r0 = rows[2*py], r1 = rows[2*py + 1];            // block-uniform: scalar loads
        const uint32_t block[4] = {separable_texel<KIND>(c0, r0), separable_texel<KIND>(c1, r0),
                                   separable_texel<KIND>(c0, r1), separable_texel<KIND>(c1, r1)};      // texel order y*2 + x (render_kernels.hpp)
        uint8_t* s = &staged[r][tid*3];
        s[0] = (uint8_t)resolve_channel_any<2>(block, a.subsample, 0);
        s[1] = (uint8_t)resolve_channel_any<2>(block, a.subsample, 8);
        s[2] = (uint8_t)resolve_channel_any<2>(block, a.subsample, 16);
    }
    (void)inside;
    __syncthreads();
    uint8_t* out = (uint8_t*)a.out + (long)frame*a.out_frame_stride;
#pragma unroll
    for (int r = 0; r < SEP_ROWS; r++) {
        const int py = blockIdx.y*SEP_ROWS + r;
        if (py < a.h) store_rgb_row(out + (long)(a.top_down ? a.h - 1 - py : py)*a.w*3, blockIdx.x*SEP_PIXELS, a.w, staged[r], tid, SEP_PIXELS, SEP_PIXELS);
    }
}

}  // namespace sf
