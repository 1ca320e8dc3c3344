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
//
// default.glsl (resources/shaders/fragment/default.glsl:1-48, the fragment of a scene that sets none) is two-dimensional — a hue
// wheel by polar angle, a ring by radius — but under the identity camera its coordinate work is not: gluv.x, the checkerboard's
// floor(uv.x*4), log2(astuv.x(1-astuv.x)) of the vignette (which factorises like the visualizer's) and `out of bounds` per column,
// the same per row. None of its two-dimensional terms decides anything a viewer could see — the hue is continuous in the angle,
// and the one branch on the radius (`circle < 0`) flips where the ring's 1/circle^2 glow has long saturated the pixel — so they
// use v_rcp/v_sqrt/v_log/v_exp and the arctangent polynomial without the IEEE divisions: within 1 LSB of the generic chain like
// every fused kernel (tests/test_gpu_pixels.py), at a third of its instructions.
#pragma once

#include "render_kernels.hpp"

namespace sf {

enum : int { SEP_BARS = 0, SEP_WAVEFORM = 1, SEP_DEFAULT = 2 };
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
    } else if (KIND == SEP_DEFAULT) {
        // { gluv, parity of floor(uv*grid/2) (default.glsl:4-8, grid = 8), log2(astuv*(1 - astuv)) (:41-43), out of bounds (camera.glsl:83) }
        // iCamera.gluv along this axis: gluv itself under the identity camera, get_camera for one coordinate under a zoomed / panned one
        bool behind = false;
        const float uv = a.identity_camera ? g : (column ? camera_along_axis<0>(u, g, a.aspect, behind) : camera_along_axis<1>(u, g, a.aspect, behind));
        const int parity = (int)::floorf(uv*8.0f/2.0f) & 1;
        e = make_float4(uv, __int_as_float(parity), __builtin_amdgcn_logf(as*(1.0f - as)), __int_as_float((column && (behind || sf::abs(g) > u.iWantAspect)) ? 1 : 0));
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

// default.glsl:10-47 for one sample from its column and row entries; `hue_shift` = 2*TAU*iTau - PI/4 (:22), per frame
__device__ __forceinline__ uint32_t default_texel(const float4 c, const float4 r, float hue_shift) {
    const float ux = c.x, uy = r.x;
    // :19 atan2(uv) in [0, 2 pi): sfmath.hpp's polynomial on min/max with hardware reciprocals (the hue is continuous in it)
    const float ax = sf::abs(ux), ay = sf::abs(uy);
    const float hi = __builtin_fmaxf(ax, ay), lo = __builtin_fminf(ax, ay);
    const float t = (hi == 0.0f) ? 0.0f : lo*__builtin_amdgcn_rcpf(hi);
    const bool upper = t > 0x1.a8279ap-2f;
    const float u = upper ? (t - 1.0f)*__builtin_amdgcn_rcpf(t + 1.0f) : t;
    const float z = u*u;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    float angle = (upper ? QUARTER_PI : 0.0f) + fmaf(p*z, u, u);
    if (ay > ax) angle = HALF_PI - angle;
    if (ux < 0.0f) angle = PI - angle;
    if (uy < 0.0f) angle = TAU - angle;
    // :22 hsv2rgb(angle + shift, 1, 1) + 0.3: the hue wheel in its continuous form
    float h = angle + hue_shift;
    h = h - TAU*::floorf(h*(1.0f/TAU));
    const float k = h*(3.0f/PI);
    const float red = clamp01(sf::abs(k - 3.0f) - 1.0f), green = clamp01(2.0f - sf::abs(k - 2.0f)), blue = clamp01(2.0f - sf::abs(k - 4.0f));
    // :25-26 the ring
    const float circle = fmaf(1.333f, __builtin_amdgcn_sqrtf(ux*ux + uy*uy), -1.0f);
    const float width = 2.0e-4f*sf::abs(__builtin_amdgcn_rcpf(circle*circle));
    // :29-33 the disc or the checkerboard
    const float base = (circle < 0.0f) ? 0.18f : (((__float_as_int(c.y) ^ __float_as_int(r.y)) & 1) ? 0.22f : 0.20f);
    // :41-43 vignette: pow(50*ax(1-ax)*ay(1-ay), 0.1) clamped
    const float vignette = clamp01(__builtin_amdgcn_exp2f(0.1f*((c.z + r.z) + 5.643856190f)));
    vec3 col = {fmaf(width, 0.3f + red, base)*vignette, fmaf(width, 0.3f + green, base)*vignette, fmaf(width, 0.3f + blue, base)*vignette};
    if (__float_as_int(c.w) != 0) col = vec3{0.15f, 0.15f, 0.15f};                               // :14-16
    return pack_rgb8(col);
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
    const float tau = a.dyn ? a.dyn[a.frame0 + frame].iTau : a.u.iTau;
    const float hue_shift = (2.0f*TAU*tau) - (PI/4.0f);               // default.glsl:22, the generic chain's operations
    (void)hue_shift;
#pragma unroll
    for (int r = 0; r < SEP_ROWS; r++) {
        const int py = blockIdx.y*SEP_ROWS + r;
        if (py >= a.h) break;
        const float4 r0 = rows[2*py], r1 = rows[2*py + 1];            // block-uniform: scalar loads
        uint32_t block[4];                                             // texel order y*2 + x (render_kernels.hpp)
        if constexpr (KIND == SEP_DEFAULT) {
            block[0] = default_texel(c0, r0, hue_shift); block[1] = default_texel(c1, r0, hue_shift);
            block[2] = default_texel(c0, r1, hue_shift); block[3] = default_texel(c1, r1, hue_shift);
        } else {
            block[0] = separable_texel<KIND>(c0, r0); block[1] = separable_texel<KIND>(c1, r0);
            block[2] = separable_texel<KIND>(c0, r1); block[3] = separable_texel<KIND>(c1, r1);
        }
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
