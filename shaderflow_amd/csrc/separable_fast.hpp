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
constexpr int SEP_PIXELS = 256;
// Output rows a block walks with its column entries in registers: 32 where the launch has blocks to spare (the entries are
// fetched once per 32 rows and the rows leave in one sweep), 8 for small launches (a single 1080p frame is 272 blocks of 32 rows)
constexpr int SEP_ROWS_LARGE = 32, SEP_ROWS_SMALL = 8;
// default.glsl is arithmetic, not stores: 16 rows (16 KB of staged rows) let eight blocks = 8 waves per SIMD live on a CU, and the
// kernel asks the compiler for that many (amdgpu_waves_per_eu: 52 registers instead of the 109 it takes when nobody asks) —
// 4 -> 8 waves per SIMD is 1.39 -> 1.17 ms per 60 frames of 4K; 8-row blocks are slower again (1.09 against 1.07 ms)
#ifndef SEP_ROWS_DEFAULT_N
#define SEP_ROWS_DEFAULT_N 16
#endif
constexpr int SEP_ROWS_DEFAULT = SEP_ROWS_DEFAULT_N;
#ifndef SEP_CHUNKS_DEFAULT
#define SEP_CHUNKS_DEFAULT 1              // walks per block where the launch has blocks to spare (see k_separable_fused)
#endif

struct SepTables {
    float4* columns;                     // [frame][wr]
    float4* rows;                        // [frame][hr]
    uint8_t* done;                       // default.glsl in two passes (k_default_quads): [frame][groups of four rows][256-pixel blocks], 1 = already written. nullptr: one pass
    int done_groups, done_blocks;
};

// astuv.y of sample row j (vertex/default.glsl:1-17 for one coordinate, glsl.hpp make_varyings) and the number of rows below a height
__device__ __forceinline__ float row_astuv(int j, int hr, float inv_hr) { return ((pixel_centre(j, hr, inv_hr)*2.0f - 1.0f) + 1.0f)/2.0f; }
__device__ __forceinline__ int rows_below(float height, int hr, float inv_hr) {
    int lo = 0, hi = hr;                           // rows [0, lo) are below, rows [hi, hr) are not (a NaN height: none is)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (row_astuv(mid, hr, inv_hr) < height) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// gluv.y of sample row j, and the rows with abs(gluv.y) < amplitude as first row | count << 16 (none for an amplitude <= 0 or NaN)
__device__ __forceinline__ float row_gluv(int j, int hr, float inv_hr) { return pixel_centre(j, hr, inv_hr)*2.0f - 1.0f; }
__device__ __forceinline__ int rows_inside(float amplitude, int hr, float inv_hr) {
    int lo = 0, hi = hr;                           // first row with gluv.y > -amplitude
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (row_gluv(mid, hr, inv_hr) > -amplitude) hi = mid; else lo = mid + 1;
    }
    const int first = lo;
    lo = 0; hi = hr;                               // first row with gluv.y >= amplitude
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (row_gluv(mid, hr, inv_hr) < amplitude) lo = mid + 1; else hi = mid;
    }
    return first | (((lo > first) ? lo - first : 0) << 16);
}

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
            // :14-17 compare astuv.y with three heights. astuv.y does not decrease with the sample row, so `astuv.y < height` is
            // `row < T(height)` with T = the number of rows below the height: three row counts per column instead of a float
            // comparison per supersample and channel (rows_below searches the very row function the row entries come from).
            const float mean = (intensity.y + intensity.x)/2.0f;
            e = make_float4(__int_as_float(rows_below(intensity.x, a.hr, a.inv_hr)), __int_as_float(rows_below(intensity.y, a.hr, a.inv_hr)),
                            __int_as_float(rows_below(mean, a.hr, a.inv_hr)), 0.4f*(intensity.x + intensity.y));
        } else {
            e = make_float4(as, 1.0f - as, 0.0f, 0.0f);
        }
    } else if (KIND == SEP_DEFAULT) {
        // { gluv, parity of floor(uv*grid/2) (default.glsl:4-8, grid = 8), this axis' factor of the vignette (:41-43), out of bounds (camera.glsl:83) }
        // pow(50*ax(1-ax)*ay(1-ay), 0.1) = (sqrt(50)*ax(1-ax))^0.1 * (sqrt(50)*ay(1-ay))^0.1: one exp2 per table entry instead of one per supersample
        // iCamera.gluv along this axis: gluv itself under the identity camera, get_camera for one coordinate under a zoomed / panned one
        bool behind = false;
        const float uv = a.identity_camera ? g : (column ? camera_along_axis<0>(u, g, a.aspect, behind) : camera_along_axis<1>(u, g, a.aspect, behind));
        const int parity = (int)::floorf(uv*8.0f/2.0f) & 1;
        const float vignette = __builtin_amdgcn_exp2f(0.1f*(__builtin_amdgcn_logf(as*(1.0f - as)) + 2.821928095f));
        if (column) {
            e = make_float4(uv, __int_as_float(parity), vignette, __int_as_float((behind || sf::abs(g) > u.iWantAspect) ? 1 : 0));
        } else {
            // .w of a row: the mean of the vignette factors of the two sample rows of its output pixel at 2x SSAA (the smooth tier of
            // k_separable_fused resolves a pixel from means: mean_ij c_i r_j = mean_i c_i * mean_j r_j)
            const int other = ((index ^ 1) < n) ? (index ^ 1) : index;
            const float ag_other = pixel_centre(other, n, a.inv_hr)*2.0f - 1.0f;
            const float as_other = (ag_other + 1.0f)/2.0f;
            const float vignette_other = __builtin_amdgcn_exp2f(0.1f*(__builtin_amdgcn_logf(as_other*(1.0f - as_other)) + 2.821928095f));
            const float mean = 0.5f*(vignette + vignette_other);
            e = make_float4(uv, __int_as_float(parity), vignette, mean);
        }
    } else {
        if (column) {
            const vec2 w = texture_xy(tex[TEX_WAVEFORM], vec2{as, 0.0f});                                          // waveform.frag:6
            // :13-15 compare abs(gluv.y) with three amplitudes. gluv.y does not decrease with the sample row, so `abs(gluv.y) < amplitude`
            // (-amplitude < gluv.y < amplitude) holds on one run of rows: its first row and its length per column and channel
            e = make_float4(__int_as_float(rows_inside(w.x, a.hr, a.inv_hr)), __int_as_float(rows_inside(w.y, a.hr, a.inv_hr)),
                            __int_as_float(rows_inside((w.x + w.y)/2.0f, a.hr, a.inv_hr)), 0.0f);
        } else {
            e = make_float4(sf::abs(g), 0.0f, 0.0f, 0.0f);                                                         // abs(gluv.y), :13
        }
    }
    (column ? t.columns + (long)frame*a.wr : t.rows + (long)frame*a.hr)[index] = e;
}

// default.glsl:19-26, the two-dimensional part of one point in two halves: the hue wheel by polar angle (as 0.3 + hsv2rgb, :22) and
// the ring `width` = 2e-4/circle² by radius; `hue_shift` = 2*TAU*iTau - PI/4 (:22), per frame
struct DefaultHue { float red, green, blue; };
struct DefaultRing { float circle, width, len; };
// the hue wheel's coordinate k = 3/PI * mod(atan2(uv) + shift, TAU) in [0, 6)
__device__ __forceinline__ float default_wheel(float ux, float uy, float hue_shift) {
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
    float h = angle + hue_shift;
    h = h - TAU*::floorf(h*(1.0f/TAU));
    return h*(3.0f/PI);
}
// :22 hsv2rgb(angle + shift, 1, 1) + 0.3: the hue wheel in its continuous form, piecewise linear in k
__device__ __forceinline__ DefaultHue default_wheel_colours(float k) {
    return DefaultHue{0.3f + clamp01(sf::abs(k - 3.0f) - 1.0f), 0.3f + clamp01(2.0f - sf::abs(k - 2.0f)), 0.3f + clamp01(2.0f - sf::abs(k - 4.0f))};
}
__device__ __forceinline__ DefaultHue default_hue(float ux, float uy, float hue_shift) { return default_wheel_colours(default_wheel(ux, uy, hue_shift)); }
// the wheel at a sample next to a point whose coordinate is known: d(angle) = (x dy - y dx)/len², exact to second order in the
// step over the radius (the callers keep that under 1/8: relative error < 1 % of a step that is itself a fraction of a hue)
__device__ __forceinline__ DefaultHue default_hue_beside(float k, float step) {
    const float moved = k + step;
    return default_wheel_colours(moved - 6.0f*::floorf(moved*(1.0f/6.0f)));
}
__device__ __forceinline__ DefaultRing default_ring(float ux, float uy) {                                           // :25-26
    DefaultRing out;
    out.len = __builtin_amdgcn_sqrtf(ux*ux + uy*uy);
    out.circle = fmaf(1.333f, out.len, -1.0f);
    out.width = 2.0e-4f*sf::abs(__builtin_amdgcn_rcpf(out.circle*out.circle));
    return out;
}
// default.glsl:29-47 for sample K of a pixel given the polar terms: `base255` = 255 * (the disc's 0.18 or the checkerboard's 0.20 /
// 0.22, :29-33), the ring added (:36, `width255` = 255*width), times the vignette clamped (:41-43, column factor * row factor): the
// unorm8 scale rides on the constants. The sample's bytes go to byte K of one register per channel (v_cvt_pk_u8_f32 writes any
// byte of its destination).
struct DefaultBytes { uint32_t red, green, blue; };
constexpr float DEFAULT_DISC = 0.18f*255.0f, DEFAULT_EVEN = 0.20f*255.0f, DEFAULT_ODD = 0.22f*255.0f;
template <int K>
__device__ __forceinline__ void default_colour(DefaultBytes& out, const DefaultHue& hue, float width255, float base255, float vignette) {
    const float ring = width255*vignette, base = base255*vignette;
    out.red = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ring, hue.red, base), K, out.red);
    out.green = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ring, hue.green, base), K, out.green);
    out.blue = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ring, hue.blue, base), K, out.blue);
}
// May several samples share ONE evaluation of the polar terms, made at the point in their middle? `reach` = the distance from that
// point to the farthest sample (in gluv units: the tables' own spacing, so zoomed cameras and small frames are measured as they
// are). Moving by d changes the ring term width*(0.3 + hue) through
//   the width:  width(len + d) = width*(1 - 2e + 3e² - ...), e = 1.333*d/circle — the linear term is applied per sample
//               (DefaultSlope), the quadratic one, 3*(1.333*reach/circle)²*width, is what sharing leaves out;
//   the hue:    width * d(hue) <= width * (3/PI) * reach/len,
// each kept under SEP_DEFAULT_TOLERANCE. Rounds 3-5 asked for 4e-5 (0.01 LSB of a channel: a sample's quantisation then flips with
// probability 2 %). Round 6: 4e-4 = 0.1 LSB per term — two terms, 0.2 LSB, plus the half LSB by which a pixel resolved from MEANS (the smooth
// tier) can differ from the mean of its samples' roundings: 0.7 < 1, so the frame still differs from the reference's by ONE LSB at most
// (`test_basic_whole_frame_4k`: ten cameras and hue shifts, whole 4K frames; at 1.2e-3 the sum is 1.1 and the test finds the pixels), and
// the per-sample band — 60 instructions per supersample on 16 % of the pixels, two thirds of the kernel's time — shrinks: 96 100 -> 110 500
// frames/s at 4K 2xSSAA (0.30 -> 0.345 of the HBM roof; 1e-4: 105 400). The tiers of k_separable_fused:
//   * FOUR rows of pixels (sixteen samples per column pair) share one evaluation where both hold — |circle| > 0.1 at 4K, 87 % of the
//     frame; the group's first row makes it, the others reuse it;
//   * else the steep 1/circle² is evaluated per sample (a square root and a reciprocal each) under the hue at the pixel's centre;
//   * on the ring itself the glow is wide enough to show the hue turning inside a pixel: the centre's wheel coordinate is moved
//     by each sample's own angle (default_hue_beside: one atan2 per pixel instead of four);
//   * next to the origin (len < 8 reach) every sample is evaluated by itself.
#ifndef SEP_DEFAULT_TOLERANCE
#define SEP_DEFAULT_TOLERANCE 4.0e-4f     // of a channel's full scale: 0.1 LSB per term (round 6; rounds 3-5: 4e-5 — see above)
#endif
__device__ __forceinline__ bool default_shares_slope(const DefaultRing& centre, float reach) {
    const float e = 1.333f*reach, c2 = centre.circle*centre.circle;
    return (3.5f*e*e*centre.width < SEP_DEFAULT_TOLERANCE*c2) && (centre.len > 8.0f*reach);
}
// d(255*width)/d(position) at the centre, along x and y: -2*1.333*width/circle * (x, y)/len
struct DefaultSlope { float x, y; };
__device__ __forceinline__ DefaultSlope default_slope(const DefaultRing& centre, float cx, float cy) {
    const float g = -2.666f*255.0f*centre.width*__builtin_amdgcn_rcpf(centre.circle*centre.len);
    return DefaultSlope{g*cx, g*cy};
}
__device__ __forceinline__ bool default_shares_hue(float width, float len, float off) { return 0.955f*width*off < SEP_DEFAULT_TOLERANCE*len; }

#ifndef SEP_DEFAULT_FLOAT_MEAN
#define SEP_DEFAULT_FLOAT_MEAN 1        // 0: every pixel of the shared tier resolved from its four quantised samples (round 3), for A/B
#endif
// which samples of the 2 x 2 block at sample row j0 are inside the runs of its two columns (bit y*2 + x, the block's texel order)
__device__ __forceinline__ bool row_in_run(int run, int j) { return (unsigned)(j - (run & 0xffff)) < ((unsigned)run >> 16); }
__device__ __forceinline__ int wave_pattern(int run0, int run1, int j0) {
    return (row_in_run(run0, j0) ? 1 : 0) | (row_in_run(run1, j0) ? 2 : 0) | (row_in_run(run0, j0 + 1) ? 4 : 0) | (row_in_run(run1, j0 + 1) ? 8 : 0);
}
// how many of the sample rows j0, j0 + 1 are among the first `below` rows (one v_med3_i32)
__device__ __forceinline__ int rows_of_pair(int below, int j0) { const int n = below - j0; return (n < 0) ? 0 : ((n > 2) ? 2 : n); }
// bars.frag:17, the blue channel of one sample as the iScreen texel's byte (bits 16-23): (below the mean ? 1 : 0) + 0.4*sum*(1 - astuv.y)
__device__ __forceinline__ uint32_t blue_texel(float below, float ramp, float one_minus_y) {
    return __builtin_amdgcn_cvt_pk_u8_f32((below + ramp*one_minus_y)*255.0f, 2u, 0u);
}

#ifndef SEP_WAVES
#define SEP_WAVES 8                     // waves per SIMD asked of the compiler (tools/variants.sh: 4 halves the occupancy, to tell latency from issue)
#endif
// what is asked is what a CU can hold: a block is four waves (one per SIMD) and SEP_ROWS KB of staged rows, so 160 KB of LDS keep
// 160/SEP_ROWS blocks — the 32-row walks live at 4 waves per SIMD and say so (round 4 asked for 8 everywhere and the compiler
// answered "final occupancy is 4" on every build)
constexpr int sep_waves(int sep_rows) {
    const int by_lds = (160*1024)/(sep_rows*SEP_PIXELS*4 + 64);
    return by_lds < SEP_WAVES ? by_lds : SEP_WAVES;
}
// S == 2. grid (ceil(w/SEP_PIXELS), ceil(h/(SEP_ROWS*CHUNKS)), frames), block SEP_PIXELS threads. A block makes CHUNKS walks of SEP_ROWS
// rows one below the other with its column entries in registers: the stores of a walk drain while the next one is computed. (A wave
// does not retire before its stores are acknowledged — microseconds under load — and a block that makes one walk spends that time
// holding its registers and LDS: default.glsl's smooth frames ran at 3.7 TB/s with their waves 87 % idle, and neither fewer
// instructions nor fewer LDS operations moved them.)
// (TWO_PASS: default.glsl after k_default_quads — the instance that reads that kernel's marks; a template argument because even
// the untaken branches on a null map cost the one-pass kernel 3 %)
template <int KIND, int SEP_ROWS, int CHUNKS = 1, bool TWO_PASS = false>
__global__ __launch_bounds__(SEP_PIXELS) __attribute__((amdgpu_waves_per_eu(sep_waves(SEP_ROWS)))) void k_separable_fused(const RenderArgs a, const SepTables t) {
    // a pixel is staged as ONE dword (red in the low byte): 64 lanes write 64 consecutive banks in one pass. (As three byte stores per
    // pixel — lanes sharing dwords — the LDS' write port bounded the smooth frames of default.glsl: 6.4 us where issue asks for 4.)
    __shared__ __attribute__((aligned(16))) uint32_t staged[SEP_ROWS][SEP_PIXELS];
    const int frame = blockIdx.z;
    const int tid = threadIdx.x;
    const int px = blockIdx.x*SEP_PIXELS + tid;
#ifdef SEP_LDS_PAD                                                     // tools/variants.sh: LDS nobody uses, to cap the blocks per CU (latency or issue?)
    __shared__ uint32_t occupancy_pad[SEP_LDS_PAD/4];
    if (a.w < 0) occupancy_pad[tid] = (uint32_t)tid;
#endif
    const float4* columns = t.columns + (long)frame*a.wr;
    const float4* rows = t.rows + (long)frame*a.hr;
    const int i0 = (2*px < a.wr) ? 2*px : a.wr - 1, i1 = (2*px + 1 < a.wr) ? 2*px + 1 : a.wr - 1;
    const float4 c0 = columns[i0], c1 = columns[i1];
    // waveform.frag: every channel of a sample is 1 or 0.2, so a pixel's byte is a function of WHICH of its four samples are inside
    // the wave: sixteen blocks, resolved once per thread block by the chain every kernel uses
    __shared__ uint8_t inside_lut[16];
    if constexpr (KIND == SEP_WAVEFORM) {
        if (tid < 16) {
            uint32_t pattern[4];
#pragma unroll
            for (int k = 0; k < 4; k++) pattern[k] = unorm8(((tid >> k) & 1) ? 1.0f : 0.2f);
            inside_lut[tid] = (uint8_t)resolve_channel_any<2>(pattern, a.subsample, 0);
        }
        __syncthreads();
    }
    const float tau = a.dyn ? a.dyn[a.frame0 + frame].iTau : a.u.iTau;
    const float hue_shift = (2.0f*TAU*tau) - (PI/4.0f);               // default.glsl:22, the generic chain's operations
    (void)hue_shift;
    const bool odd_column0 = (__float_as_int(c0.y) & 1) != 0, odd_column1 = (__float_as_int(c1.y) & 1) != 0;   // default.glsl only
    const bool outside = (__float_as_int(c0.w) | __float_as_int(c1.w)) != 0;
    const float cx = 0.5f*(c0.x + c1.x);
    // smooth tier of default.glsl: the pixel column's means (see the row loop)
    // (few registers across the walk — the kernel lives at 64: the odd rows' means are (0.20 + 0.22)*255 times the column mean minus the
    // even rows', what a group needs beyond that it derives when it begins)
    float smooth_cz = 0.0f, smooth_g0 = 0.0f, smooth_b0 = 0.0f;
    int smooth_until = 0;                                              // wave-uniform: rows [r, smooth_until) of the walk are smooth for every lane of the wave
    const bool wave_one_square = __builtin_amdgcn_ballot_w64(odd_column0 == odd_column1) == __builtin_amdgcn_ballot_w64(true);
    if constexpr (KIND == SEP_DEFAULT) {
        // the checkerboard under a sample: (column parity ^ row parity) ? 0.22 : 0.20 (default.glsl:4-8)
        const float even_rows0 = odd_column0 ? DEFAULT_ODD : DEFAULT_EVEN, even_rows1 = odd_column1 ? DEFAULT_ODD : DEFAULT_EVEN;
        smooth_cz = 0.5f*(c0.z + c1.z);
        smooth_g0 = 0.5f*fmaf(c0.z, even_rows0, c1.z*even_rows1);
        smooth_b0 = 0.5f*(even_rows0 + even_rows1);
    }
    (void)smooth_cz; (void)smooth_g0; (void)smooth_b0; (void)smooth_until; (void)wave_one_square;
    float group_wheel = 0.0f; int wheel_until = 0; (void)group_wheel; (void)wheel_until;
    int shared_until = 0; bool attempted = false; (void)attempted; DefaultRing group_ring = {}; DefaultHue group_hue = {}; DefaultSlope group_slope = {}; float group_y = 0.0f;
    (void)odd_column0; (void)odd_column1; (void)outside; (void)cx; (void)shared_until; (void)group_ring; (void)group_hue; (void)group_slope; (void)group_y;
#pragma unroll 1
    for (int chunk = 0; chunk < CHUNKS; chunk++) {
        const int block_row = blockIdx.y*CHUNKS + chunk;               // which walk of the frame
        if (block_row*SEP_ROWS >= a.h) break;
        if (chunk) {
            __syncthreads();                                           // the previous walk's rows have left the LDS
            smooth_until = 0; shared_until = 0; attempted = false; wheel_until = 0;
        }
        // default.glsl in two passes: the groups of four rows k_default_quads has already written (block-uniform: a block's 256 pixels are
        // one wave of that kernel)
        unsigned done_mask = 0u;
        if constexpr (KIND == SEP_DEFAULT && TWO_PASS) {
            if (t.done) {
                const uint8_t* done = t.done + ((long)frame*t.done_groups + (block_row*SEP_ROWS)/4)*t.done_blocks + blockIdx.x;
#pragma unroll
                for (int g = 0; g < SEP_ROWS/4; g++)
                    if ((int)block_row*SEP_ROWS + 4*g < a.h && done[(long)g*t.done_blocks]) done_mask |= 1u << g;
                done_mask = __builtin_amdgcn_readfirstlane(done_mask);
                const int groups_here = (min((int)block_row*SEP_ROWS + SEP_ROWS, a.h) - (int)block_row*SEP_ROWS + 3)/4;
                if (done_mask == (1u << groups_here) - 1u) continue;    // nothing left of this walk (no barrier is skipped by only some threads: the mask is the block's)
            }
        }
        if constexpr (KIND == SEP_DEFAULT) {
            // default.glsl: the walk in groups of four rows. (A row of the tiers below is ~3 KB of code: it exists once, in a loop.)
            static_assert(SEP_ROWS % 4 == 0, "the walk is made of groups of four rows");
            auto stage = [&](int r, uint32_t rgb) {
                staged[r][tid] = rgb;
            };
            // one row of a smooth group from its two row entries (see the group's prologue): eleven vector operations and nothing
            // scalar — a wave issues one instruction of ANY kind per turn, so the selects and branches on the rows' parity that used to
            // sit here cost as much as the arithmetic (347 scalar against 470 vector instructions per wave)
            // (`offset` = the ring's width at gluv.y = 0 by the group's slope: w(y) = offset + slope.y*y; `ground_of_rows`, `base_of_rows` =
            // the means of the group's rows: the disc's, or the checkerboard's by the parity its eight sample rows share)
            auto smooth_row = [&](const float4 r0, const float4 r1, const bool one_square, const float ground_of_rows, const float base_of_rows, const float half_slope, const float offset) -> uint32_t {
#ifdef SEP_STUB_SMOOTH                                                 // tools/variants.sh: what do the smooth rows' operations cost?
                return __float_as_uint(r0.x + smooth_cz) & 0xffffffu;
#endif
                const float vignette = clamp01(smooth_cz*r0.w);
                const float ring = fmaf(half_slope, r0.x, fmaf(half_slope, r1.x, offset))*vignette;
                const float ground = one_square ? base_of_rows*vignette                       // min(r*c*B, B) = B*min(r*c, 1)
                                                : __builtin_fminf(r0.w*ground_of_rows, base_of_rows);
                uint32_t rgb = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ring, group_hue.red, ground), 0u, 0u);
                rgb = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ring, group_hue.green, ground), 1u, rgb);
                return __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ring, group_hue.blue, ground), 2u, rgb);
            };
            // one row by the tiers that look at every pixel by itself
            auto general_row = [&](int r, const float4 r0, const float4 r1) -> uint32_t {
                uint32_t rgb = 0u;
                // the hue wheel's coordinate at the pixel's centre: from the group's, or by itself
                auto wheel_at = [&](float cy) -> float {
                    if (r < wheel_until) {
                        const float moved = fmaf((3.0f/PI)*cx*__builtin_amdgcn_rcpf(group_ring.len*group_ring.len), cy - group_y, group_wheel);
                        return moved - 6.0f*::floorf(moved*(1.0f/6.0f));
                    }
                    return default_wheel(cx, cy, hue_shift);
                };
                const bool group_shares = r < shared_until;
                // the sample's checkerboard colour by the parities of its column and row; the row's side is block-uniform (scalar selects)
                const float even0 = (__float_as_int(r0.y) & 1) ? DEFAULT_ODD : DEFAULT_EVEN, odd0 = (__float_as_int(r0.y) & 1) ? DEFAULT_EVEN : DEFAULT_ODD;
                const float even1 = (__float_as_int(r1.y) & 1) ? DEFAULT_ODD : DEFAULT_EVEN, odd1 = (__float_as_int(r1.y) & 1) ? DEFAULT_EVEN : DEFAULT_ODD;
                // (functions, not arrays: evaluated where a tier uses them — the tier that resolves from means does not)
                auto board_of = [&](int k) -> float { return (k & 1 ? odd_column1 : odd_column0) ? (k & 2 ? odd1 : odd0) : (k & 2 ? even1 : even0); };
                auto vignette_of = [&](int k) -> float { return clamp01((k & 1 ? c1.z : c0.z)*(k & 2 ? r1.z : r0.z)); };
                const float cy = 0.5f*(r0.x + r1.x);
                const float off = 0.5f*(sf::abs(c1.x - c0.x) + sf::abs(r1.x - r0.x));
                DefaultRing centre = group_ring;
                DefaultHue hue = group_hue;
                DefaultSlope slope = group_slope;
                float slope_y = group_y;
                bool shared = group_shares;
                if (!group_shares) {
                    centre = default_ring(cx, cy);
                    // the same test for the pixel alone: its reach is a seventh of the group's, so half of the band that four rows cannot
                    // share (0.05 < |circle| < 0.1 at 4K) is served by one evaluation per pixel instead of four
                    shared = default_shares_slope(centre, off) && default_shares_hue(1.5f*centre.width, centre.len, off);
#ifdef SEP_STUB_UNSHARED                                               // tools/variants.sh: what does the per-sample tier cost? (wrong pixels in the band)
                    shared = true;
#endif
                    if (shared) { hue = default_wheel_colours(wheel_at(cy)); slope = default_slope(centre, cx, cy); slope_y = cy; }
                }
                DefaultBytes bytes = {0u, 0u, 0u};
                bool meaned = false;
                if (shared) {                                             // one ring and one hue for the samples, the ring's width to first order
                    const float lower = fmaf(slope.y, r0.x - slope_y, centre.width*255.0f), upper = fmaf(slope.y, r1.x - slope_y, centre.width*255.0f);
                    const float side = slope.x*(0.5f*(c1.x - c0.x));
                    const bool disc = centre.circle < 0.0f;
#if SEP_DEFAULT_FLOAT_MEAN
                    // the pixel from MEANS, as a smooth row does (see the group's prologue), with the ring's width at the pixel's centre:
                    // mean_k vig_k*(base_k + w_k*hue) = min(r*G, B) + min(c*r, 1)*w(centre)*hue up to second-order terms — twelve
                    // operations, no per-sample board or vignette. Within one of the reference like every float mean (no sample
                    // saturates in a shared tier: |circle| > 0.05 keeps the ring term under 0.1).
                    if (!outside && (((__float_as_int(r0.y) ^ __float_as_int(r1.y)) & 1) == 0)) {
                        const bool odd_rows = (__float_as_int(r0.y) & 1) != 0;
                        const float vignette = clamp01(smooth_cz*r0.w);
                        const float ring = fmaf(slope.y, cy - slope_y, centre.width*255.0f)*vignette;
                        const float ground = __builtin_fminf(r0.w*(disc ? DEFAULT_DISC*smooth_cz : (odd_rows ? (DEFAULT_EVEN + DEFAULT_ODD)*smooth_cz - smooth_g0 : smooth_g0)), disc ? DEFAULT_DISC : (odd_rows ? (DEFAULT_EVEN + DEFAULT_ODD) - smooth_b0 : smooth_b0));
                        rgb = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ring, hue.red, ground), 0u, 0u);
                        rgb = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ring, hue.green, ground), 1u, rgb);
                        rgb = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ring, hue.blue, ground), 2u, rgb);
                        meaned = true;
                    }
#endif
                    if (!meaned) {
                    default_colour<0>(bytes, hue, lower - side, disc ? DEFAULT_DISC : board_of(0), vignette_of(0));
                    default_colour<1>(bytes, hue, lower + side, disc ? DEFAULT_DISC : board_of(1), vignette_of(1));
                    default_colour<2>(bytes, hue, upper - side, disc ? DEFAULT_DISC : board_of(2), vignette_of(2));
                    default_colour<3>(bytes, hue, upper + side, disc ? DEFAULT_DISC : board_of(3), vignette_of(3));
                    }
                } else {                                                  // the ring per sample; the hue at the centre if it may be, else per sample
                    const float ux[4] = {c0.x, c1.x, c0.x, c1.x}, uy[4] = {r0.x, r0.x, r1.x, r1.x};
                    DefaultRing ring[4];
                    float widest = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; k++) { ring[k] = default_ring(ux[k], uy[k]); widest = __builtin_fmaxf(widest, ring[k].width); }
                    float width255[4], base255[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) { width255[k] = ring[k].width*255.0f; base255[k] = (ring[k].circle < 0.0f) ? DEFAULT_DISC : board_of(k); }
                    if (centre.len > 8.0f*off) {
                        const float wheel = wheel_at(cy);
                        if (default_shares_hue(widest, centre.len, off)) {
                            hue = default_wheel_colours(wheel);
                            default_colour<0>(bytes, hue, width255[0], base255[0], vignette_of(0)); default_colour<1>(bytes, hue, width255[1], base255[1], vignette_of(1));
                            default_colour<2>(bytes, hue, width255[2], base255[2], vignette_of(2)); default_colour<3>(bytes, hue, width255[3], base255[3], vignette_of(3));
                        } else {
                            // on the ring the glow is wide enough to show the hue turning inside a pixel: the centre's wheel coordinate
                            // moved by each sample's own angle (one atan2 per pixel instead of four)
                            const float turn = (3.0f/PI)*__builtin_amdgcn_rcpf(centre.len*centre.len);
                            const float along = cx*(0.5f*(r1.x - r0.x))*turn, across = cy*(0.5f*(c1.x - c0.x))*turn;   // x dy, y dx
                            default_colour<0>(bytes, default_hue_beside(wheel, across - along), width255[0], base255[0], vignette_of(0));
                            default_colour<1>(bytes, default_hue_beside(wheel, -across - along), width255[1], base255[1], vignette_of(1));
                            default_colour<2>(bytes, default_hue_beside(wheel, across + along), width255[2], base255[2], vignette_of(2));
                            default_colour<3>(bytes, default_hue_beside(wheel, along - across), width255[3], base255[3], vignette_of(3));
                        }
                    } else {                                              // next to the origin: every sample by itself
                        default_colour<0>(bytes, default_hue(ux[0], uy[0], hue_shift), width255[0], base255[0], vignette_of(0));
                        default_colour<1>(bytes, default_hue(ux[1], uy[1], hue_shift), width255[1], base255[1], vignette_of(1));
                        default_colour<2>(bytes, default_hue(ux[2], uy[2], hue_shift), width255[2], base255[2], vignette_of(2));
                        default_colour<3>(bytes, default_hue(ux[3], uy[3], hue_shift), width255[3], base255[3], vignette_of(3));
                    }
                }
                if (outside) {                                            // camera.glsl:83 / default.glsl:14-16, per column: 0.15 grey = byte 38
                    const uint32_t keep = ((__float_as_int(c0.w) != 0) ? 0u : 0x00ff00ffu) | ((__float_as_int(c1.w) != 0) ? 0u : 0xff00ff00u);
                    const uint32_t grey = 0x26262626u & ~keep;
                    bytes.red = (bytes.red & keep) | grey; bytes.green = (bytes.green & keep) | grey; bytes.blue = (bytes.blue & keep) | grey;
                }
                // final.glsl's mean of the four RGBA8 texels as an INTEGER mean per channel, (sum + 2) >> 2 with the sum of a register's
                // four bytes from v_sad_u8: what resolve_channel's float chain gives except on ties (sum = 2 mod 4), which its rounding
                // noise decides either way — 1 LSB, like every approximation of this kernel (8 instructions instead of 60)
                if (!meaned)
                    rgb = (__builtin_amdgcn_sad_u8(bytes.red, 0u, 2u) >> 2) | ((__builtin_amdgcn_sad_u8(bytes.green, 0u, 2u) >> 2) << 8)
                        | ((__builtin_amdgcn_sad_u8(bytes.blue, 0u, 2u) >> 2) << 16);
                return rgb;
            };
#pragma unroll 1
            for (int r = 0; r < SEP_ROWS; r += 4) {
                const int py = block_row*SEP_ROWS + r;
                if (py >= a.h) break;
                if (TWO_PASS && ((done_mask >> (r >> 2)) & 1u)) continue;  // written by k_default_quads
                const float4 r0 = rows[2*py], r1 = rows[2*py + 1];        // block-uniform: scalar loads
                // rows in groups of four: the group's first row evaluates the polar terms at the point in the middle of its sixteen samples
                // and, where default_shares_slope / default_shares_hue allow, the four rows use them (`shared_until` = the first block row
                // they no longer serve)
                if (r >= smooth_until) {
                    // THE SMOOTH TIER (round 4). Where every lane of the wave may share one evaluation of the polar terms, the pixel is
                    // resolved from MEANS in float: mean_k vig_k*(base_k + w_k*hue) over its four samples k = (column i, row j) with
                    // vig_k = min(c_i*r_j, 1) is, up to terms far below the quantisation step,
                    //     min(mean_j r_j * mean_i c_i*base_i, mean_i base_i)  +  min(mean_i c_i * mean_j r_j, 1) * w(pixel centre) * hue
                    // — the products factorise exactly, the clamp differs only where some of the four products exceed 1 and others do not
                    // (the middle of the frame, where they differ by 1e-4), the covariance of vig and w inside a pixel is second order
                    // (< 0.005 LSB at the frame's edge). Column means live in registers for the walk, row means in the row table (.w):
                    // ≈ 20 operations per pixel. Against the reference, which quantises each sample: the mean of four roundings is within
                    // 1/2 of the mean, so the bytes are roundings of numbers within 1/2 (+ 0.03) of each other — at most ONE apart, the
                    // bound of every fused kernel (no sample saturates here: |circle| > 0.1 keeps the ring term under 0.03).
                    // One evaluation serves the WHOLE walk of a full block when the test holds over its sixteen rows (|circle| > 0.3 at 4K:
                    // two thirds of the frame), else four rows at a time; rows whose two sample rows sit in different squares of the
                    // checkerboard, and waves with a lane that fails, take the tiers below.
                    bool whole_walk = false;
                    if (r == 0 && (int)block_row*SEP_ROWS + SEP_ROWS <= a.h && SEP_ROWS > 4) {
                        const float cy = 0.5f*(rows[2*py + SEP_ROWS - 1].x + rows[2*py + SEP_ROWS].x);
                        const float reach = 0.5f*sf::abs(c1.x - c0.x) + ((float)SEP_ROWS - 0.5f)*sf::abs(r1.x - r0.x);
                        group_ring = default_ring(cx, cy);
                        const bool fine = !outside && default_shares_slope(group_ring, reach) && default_shares_hue(1.5f*group_ring.width, group_ring.len, reach);
                        if (__builtin_amdgcn_ballot_w64(fine) == __builtin_amdgcn_ballot_w64(true)) {
                            group_hue = default_hue(cx, cy, hue_shift); group_slope = default_slope(group_ring, cx, cy); group_y = cy;
                            shared_until = SEP_ROWS; smooth_until = SEP_ROWS; whole_walk = true;
                        }
                    }
                    if (!whole_walk) {
                        attempted = py + 3 < a.h;
                        bool fine = false;
                        if (attempted) {
                            const float cy = 0.5f*(rows[2*py + 3].x + rows[2*py + 4].x);
                            const float reach = 0.5f*sf::abs(c1.x - c0.x) + 3.5f*sf::abs(r1.x - r0.x);
                            group_ring = default_ring(cx, cy);
                            if (default_shares_slope(group_ring, reach) && default_shares_hue(1.5f*group_ring.width, group_ring.len, reach)) {
                                group_hue = default_hue(cx, cy, hue_shift); group_slope = default_slope(group_ring, cx, cy); group_y = cy; shared_until = r + 4;
                                fine = !outside;
                            }
                        }
                        if (__builtin_amdgcn_ballot_w64(fine) == __builtin_amdgcn_ballot_w64(true)) smooth_until = r + 4;
                        else if (attempted) {
                            // the group's rows go through the tiers below, pixel by pixel: ONE arctangent for the four rows of a lane,
                            // at the group's middle; a row moves the wheel by its own angle x*dy/len² (default_hue_beside: second order
                            // in dy/len, < 1e-5 of the wheel where the ring lives), unless the lane sits next to the origin
                            const float cy = 0.5f*(rows[2*py + 3].x + rows[2*py + 4].x);
                            const float reach = 0.5f*sf::abs(c1.x - c0.x) + 3.5f*sf::abs(r1.x - r0.x);
                            group_wheel = default_wheel(cx, cy, hue_shift);
                            group_y = cy;
                            wheel_until = (group_ring.len > 16.0f*reach) ? r + 4 : 0;
                        }
                    }
                }
                if (r < smooth_until && py + 3 < a.h) {
                    // the group's eight row entries in one fetch (a fetch per row is a round trip to the L2 in front of fifteen operations)
                    float4 e[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) e[k] = rows[2*py + k];
                    // (the eight sample rows in one square of the checkerboard's rows — 66 groups in 67 at 4K; the others go below)
                    int mixed = 0;
#pragma unroll
                    for (int k = 1; k < 8; k++) mixed |= __float_as_int(e[k].y) ^ __float_as_int(e[0].y);
                    if ((mixed & 1) == 0) {
                        const bool odd_rows = (__float_as_int(e[0].y) & 1) != 0;
                        const bool disc = group_ring.circle < 0.0f;
                        const float ground_of_rows = disc ? DEFAULT_DISC*smooth_cz : (odd_rows ? (DEFAULT_EVEN + DEFAULT_ODD)*smooth_cz - smooth_g0 : smooth_g0);
                        const float base_of_rows = disc ? DEFAULT_DISC : (odd_rows ? (DEFAULT_EVEN + DEFAULT_ODD) - smooth_b0 : smooth_b0);
                        const float half_slope = 0.5f*group_slope.y, offset = fmaf(-group_slope.y, group_y, group_ring.width*255.0f);
                        if (wave_one_square) {                          // wave-uniform: both sample columns of every lane in one square of the checkerboard's columns
#pragma unroll
                            for (int k = 0; k < 4; k++) stage(r + k, smooth_row(e[2*k], e[2*k + 1], true, ground_of_rows, base_of_rows, half_slope, offset));
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; k++) stage(r + k, smooth_row(e[2*k], e[2*k + 1], false, ground_of_rows, base_of_rows, half_slope, offset));
                        }
                        continue;
                    }
                }
#pragma unroll 1
                for (int k = 0; k < 4; k++) {
                    if (py + k >= a.h) break;
                    const float4 q0 = rows[2*(py + k)], q1 = rows[2*(py + k) + 1];
                    stage(r + k, general_row(r + k, q0, q1));
                }
            }
        } else {
#pragma unroll
        for (int r = 0; r < SEP_ROWS; r++) {
            const int py = block_row*SEP_ROWS + r;
            if (py >= a.h) break;
            const float4 r0 = rows[2*py], r1 = rows[2*py + 1];            // block-uniform: scalar loads
            uint32_t block[4];                                             // texel order y*2 + x (render_kernels.hpp)
            uint32_t rgb;                                                  // the output pixel, red in the low byte
            if constexpr (KIND == SEP_BARS) {
                // red and green are 0 or 1 per supersample: the resolve of such a block is exact in every step — the sum of the four
                // texels/255 is their count, count/4 is exact, and the unorm8 write of it is RN(count*63.75) for either kernel size
                // (the 1-tap kernel's weights are 0.25 each) — so the byte is a function of how many of the pixel's four samples lie
                // below the column's height: min(max(T - first row, 0), 2) per column.
                const int j0 = 2*py;
                const int red = rows_of_pair(__float_as_int(c0.x), j0) + rows_of_pair(__float_as_int(c1.x), j0);
                const int green = rows_of_pair(__float_as_int(c0.y), j0) + rows_of_pair(__float_as_int(c1.y), j0);
                // blue adds the ramp 0.4*(sum)*(1 - astuv.y) (bars.frag:17): the generic chain on that channel alone
                const int t0 = __float_as_int(c0.z), t1 = __float_as_int(c1.z);
                block[0] = blue_texel((j0 < t0) ? 1.0f : 0.0f, c0.w, r0.y); block[1] = blue_texel((j0 < t1) ? 1.0f : 0.0f, c1.w, r0.y);
                block[2] = blue_texel((j0 + 1 < t0) ? 1.0f : 0.0f, c0.w, r1.y); block[3] = blue_texel((j0 + 1 < t1) ? 1.0f : 0.0f, c1.w, r1.y);
                rgb = __builtin_amdgcn_cvt_pk_u8_f32((float)red*63.75f, 0u, resolve_channel_any<2>(block, a.subsample, 16) << 16);
                rgb = __builtin_amdgcn_cvt_pk_u8_f32((float)green*63.75f, 1u, rgb);
            } else if constexpr (KIND == SEP_WAVEFORM) {
                const int j0 = 2*py;
                rgb = (uint32_t)inside_lut[wave_pattern(__float_as_int(c0.x), __float_as_int(c1.x), j0)]
                    | ((uint32_t)inside_lut[wave_pattern(__float_as_int(c0.y), __float_as_int(c1.y), j0)] << 8)
                    | ((uint32_t)inside_lut[wave_pattern(__float_as_int(c0.z), __float_as_int(c1.z), j0)] << 16);
            }
            staged[r][tid] = rgb;
        }
        }
        __syncthreads();
        uint8_t* out = (uint8_t*)a.out + (long)frame*a.out_frame_stride;
        // a full-width block of a frame whose rows are whole dwords: a lane packs four staged pixels into their twelve bytes (three
        // v_perm_b32) and stores them as one non-temporal `global_store_dwordx3`; a wave writes 768 contiguous bytes of one row per pass.
        // Wave `v` takes rows v, v + 4, ...: the row and its address are SCALAR (as a function of the thread index the compiler spent
        // twenty vector operations per pass on 64-bit products — a sixth of a smooth wave's instructions).
        const int x0 = blockIdx.x*SEP_PIXELS;
        if (x0 + SEP_PIXELS <= a.w && (a.w & 3) == 0 && ((uintptr_t)out & 3) == 0 && (a.out_frame_stride & 3) == 0) {
            const int rows_here = min(SEP_ROWS, a.h - (int)block_row*SEP_ROWS);
            typedef uint32_t Triple __attribute__((ext_vector_type(3), aligned(4)));
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
            uint8_t* const block_out = out + (long)x0*3;                  // block-uniform
            const uint32_t lane_bytes = (uint32_t)lane*12u;
#pragma unroll
            for (int i = 0; i < SEP_ROWS/4; i++) {
                const int r = wave + 4*i;
                if (r < rows_here && !(TWO_PASS && ((done_mask >> (r >> 2)) & 1u))) {
                    const int py = block_row*SEP_ROWS + r;
                    const uint4 p = *(const uint4*)&staged[r][4*lane];
                    uint8_t* row = block_out + (long)(a.top_down ? a.h - 1 - py : py)*a.w*3;
#ifdef SEP_STUB_STORES                                                 // tools/variants.sh: what do the stores cost? (one lane in 64 stores)
                    if (lane == (p.x & 63u) + 64u) continue;
                    if (lane != 0) continue;
#endif
                    __builtin_nontemporal_store(Triple{__builtin_amdgcn_perm(p.y, p.x, 0x04020100u), __builtin_amdgcn_perm(p.z, p.y, 0x05040201u),
                                                       __builtin_amdgcn_perm(p.w, p.z, 0x06050402u)}, (Triple*)(row + lane_bytes));
                }
            }
            continue;
        }
        const int pixels_here = min(SEP_PIXELS, a.w - x0);
#pragma unroll 1
        for (int r = 0; r < SEP_ROWS; r++) {
            const int py = block_row*SEP_ROWS + r;
            if (py < a.h && tid < pixels_here && !(TWO_PASS && ((done_mask >> (r >> 2)) & 1u))) {
                uint8_t* pixel = out + (long)(a.top_down ? a.h - 1 - py : py)*a.w*3 + (long)(x0 + tid)*3;
                const uint32_t rgb = staged[r][tid];
                pixel[0] = (uint8_t)rgb; pixel[1] = (uint8_t)(rgb >> 8); pixel[2] = (uint8_t)(rgb >> 16);
            }
        }
    }
}

// ---- default.glsl's smooth tier, four pixels per lane (round 5) ---------------------------------------------------------------------
// k_separable_fused<default> gives a lane ONE pixel column: a row of a wave is 192 bytes that no lane can store by itself, so every
// row is staged in LDS, read back four pixels at a time, permuted and stored — and of the 470 vector + 347 scalar instructions a smooth
// wave spends on 1 024 pixels, 176 are the fragment's arithmetic (profiles/r04_basic_tiers.txt: a frame that is smooth everywhere
// runs at 0.50 of the HBM roof with the VALU 97 % active). Here a lane owns FOUR adjacent pixels = 12 bytes = one non-temporal
// `global_store_dwordx3` straight from registers (v_cvt_pk_u8_f32 puts a byte anywhere in a dword: the three dwords of four RGB
// pixels are built in place); a wave covers 256 pixels — exactly a block of k_separable_fused — and walks DQ_ROWS rows. It serves the
// SMOOTH tier only, with that kernel's arithmetic (same evaluation points, same formula: see its smooth_row): groups of four rows where
// the test holds for all 256 pixels are written and marked in `t.done`; k_separable_fused then runs over the frame as before and
// skips what is marked (the band, 16 % of the wave-rows at 4K, keeps its per-pixel tiers and its own one-pixel lanes).
// MEASURED AND NOT THE DEFAULT (profiles/r05_basic_quads.txt): a frame that is smooth everywhere takes 6.3 us through this kernel
// and 6.4 through the staged one — the staging was not the cost — and at the Basic scene's zoom the second pass' skipping makes the
// pair slower than the one pass (84 400 against 92 300 frames/s). Kept behind SHADERFLOW_DEFAULT_QUADS=1 with its parity test.
constexpr int DQ_ROWS = 16;                                            // rows of a walk (= SEP_ROWS_DEFAULT: the whole-walk evaluation point is the same)
#ifndef DQ_WALKS
#define DQ_WALKS 4                                                     // walks a wave makes one below the other with its columns in registers (a wave that ends waits for its stores' acknowledgements)
#endif
#ifndef DQ_WAVES
#define DQ_WAVES 4                                                     // 128 registers, no spill: 5 waves spill 15, 6 spill 33 and lose a quarter (profiles/r05_basic_quads.txt)
#endif
struct DefaultQuadPixel { float cx, half_reach, cz, g0, b0; };        // a pixel column: gluv.x at its centre, half the sample spacing, the column means of the smooth tier
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DQ_WAVES))) void k_default_quads(const RenderArgs a, const SepTables t) {
    const int frame = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int y_wave = (blockIdx.y*4 + wave)*DQ_ROWS*DQ_WALKS;
    if (y_wave >= a.h) return;
    const int bx = blockIdx.x;
    const bool full_block = bx*256 + 256 <= a.w && (a.w & 3) == 0 && (a.out_frame_stride & 3) == 0 && (((uintptr_t)a.out) & 3) == 0;
    if (!full_block) {                                                 // a partial block of columns: the other kernel's
        const int groups = (min(y_wave + DQ_ROWS*DQ_WALKS, a.h) - y_wave + 3)/4;
        uint8_t* done = t.done + ((long)frame*t.done_groups + y_wave/4)*t.done_blocks + bx;
        for (int g = lane; g < groups; g += 64) done[(long)g*t.done_blocks] = 0;
        return;
    }
    const int px0 = bx*256 + lane*4;
    const float4* columns = t.columns + (long)frame*a.wr + 2*px0;
    const float4* rows = t.rows + (long)frame*a.hr;
    const float tau = a.dyn ? a.dyn[a.frame0 + frame].iTau : a.u.iTau;
    const float hue_shift = (2.0f*TAU*tau) - (PI/4.0f);               // default.glsl:22
    DefaultQuadPixel pixel[4];
    bool outside = false;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const float4 c0 = columns[2*p], c1 = columns[2*p + 1];
        const bool odd0 = (__float_as_int(c0.y) & 1) != 0, odd1 = (__float_as_int(c1.y) & 1) != 0;
        const float even_rows0 = odd0 ? DEFAULT_ODD : DEFAULT_EVEN, even_rows1 = odd1 ? DEFAULT_ODD : DEFAULT_EVEN;
        pixel[p].cx = 0.5f*(c0.x + c1.x);
        pixel[p].half_reach = 0.5f*sf::abs(c1.x - c0.x);
        pixel[p].cz = 0.5f*(c0.z + c1.z);
        pixel[p].g0 = 0.5f*fmaf(c0.z, even_rows0, c1.z*even_rows1);
        pixel[p].b0 = 0.5f*(even_rows0 + even_rows1);
        outside = outside || ((__float_as_int(c0.w) | __float_as_int(c1.w)) != 0);
    }
    uint8_t* out = (uint8_t*)a.out + (long)frame*a.out_frame_stride + (long)px0*3;
    const long pitch = (long)a.w*3;
    const float row_step = sf::abs(rows[2*y_wave + 1].x - rows[2*y_wave].x);
    const unsigned long long everybody = __builtin_amdgcn_ballot_w64(true);

    // one evaluation of the polar terms per pixel, at (its column's centre, cy): may `rows_reach` rows around cy share it? (k_separable_fused)
    DefaultRing ring[4]; DefaultHue hue[4]; float slope_y[4], group_y = 0.0f;
    auto evaluate = [&](float cy, float rows_reach) -> bool {
        bool fine = !outside;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const float reach = pixel[p].half_reach + rows_reach*row_step;
            ring[p] = default_ring(pixel[p].cx, cy);
            fine = fine && default_shares_slope(ring[p], reach) && default_shares_hue(1.5f*ring[p].width, ring[p].len, reach);
        }
        if (__builtin_amdgcn_ballot_w64(fine) != everybody) return false;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            hue[p] = default_hue(pixel[p].cx, cy, hue_shift);
            slope_y[p] = default_slope(ring[p], pixel[p].cx, cy).y;
        }
        group_y = cy;
        return true;
    };
#pragma unroll 1
    for (int walk = 0; walk < DQ_WALKS; walk++) {
    const int y_first = y_wave + walk*DQ_ROWS;
    if (y_first >= a.h) break;
    uint8_t* done = t.done + ((long)frame*t.done_groups + y_first/4)*t.done_blocks + bx;
    const int groups_here = (min(y_first + DQ_ROWS, a.h) - y_first + 3)/4;
    const bool whole_walk = (y_first + DQ_ROWS <= a.h) && evaluate(0.5f*(rows[2*y_first + DQ_ROWS - 1].x + rows[2*y_first + DQ_ROWS].x), (float)DQ_ROWS - 0.5f);
#pragma unroll 1
    for (int g = 0; g < groups_here; g++) {
        const int py = y_first + 4*g;
        bool smooth = py + 3 < a.h;
        if (smooth && !whole_walk) smooth = evaluate(0.5f*(rows[2*py + 3].x + rows[2*py + 4].x), 3.5f);
        float4 e[8];
        if (smooth) {
            int mixed = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) e[k] = rows[2*py + k];        // wave-uniform: scalar loads
#pragma unroll
            for (int k = 1; k < 8; k++) mixed |= __float_as_int(e[k].y) ^ __float_as_int(e[0].y);
            smooth = (mixed & 1) == 0;                                 // the eight sample rows in one square of the checkerboard's rows
        }
        if (lane == 0) done[(long)g*t.done_blocks] = smooth ? 1 : 0;
        if (!smooth) continue;
        const bool odd_rows = (__float_as_int(e[0].y) & 1) != 0;
        float ground_of_rows[4], base_of_rows[4], half_slope[4], offset[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const bool disc = ring[p].circle < 0.0f;
            ground_of_rows[p] = disc ? DEFAULT_DISC*pixel[p].cz : (odd_rows ? (DEFAULT_EVEN + DEFAULT_ODD)*pixel[p].cz - pixel[p].g0 : pixel[p].g0);
            base_of_rows[p] = disc ? DEFAULT_DISC : (odd_rows ? (DEFAULT_EVEN + DEFAULT_ODD) - pixel[p].b0 : pixel[p].b0);
            half_slope[p] = 0.5f*slope_y[p];
            offset[p] = fmaf(-slope_y[p], group_y, ring[p].width*255.0f);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // the row's three values in VECTOR registers: a full-rate form with a scalar source issues at half rate (profiles/r05_ubench_valu_sgpr.txt)
            float y0, y1, rw;
            asm("v_mov_b32 %0, %1" : "=v"(y0) : "s"(e[2*k].x));
            asm("v_mov_b32 %0, %1" : "=v"(y1) : "s"(e[2*k + 1].x));
            asm("v_mov_b32 %0, %1" : "=v"(rw) : "s"(e[2*k].w));
            uint32_t d0 = 0u, d1 = 0u, d2 = 0u;                       // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
            float red[4], green[4], blue[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const float vignette = clamp01(pixel[p].cz*rw);
                const float glow = fmaf(half_slope[p], y0, fmaf(half_slope[p], y1, offset[p]))*vignette;
                const float ground = __builtin_fminf(rw*ground_of_rows[p], base_of_rows[p]);
                red[p] = fmaf(glow, hue[p].red, ground); green[p] = fmaf(glow, hue[p].green, ground); blue[p] = fmaf(glow, hue[p].blue, ground);
            }
            d0 = __builtin_amdgcn_cvt_pk_u8_f32(red[0], 0u, d0); d0 = __builtin_amdgcn_cvt_pk_u8_f32(green[0], 1u, d0); d0 = __builtin_amdgcn_cvt_pk_u8_f32(blue[0], 2u, d0);
            d0 = __builtin_amdgcn_cvt_pk_u8_f32(red[1], 3u, d0); d1 = __builtin_amdgcn_cvt_pk_u8_f32(green[1], 0u, d1); d1 = __builtin_amdgcn_cvt_pk_u8_f32(blue[1], 1u, d1);
            d1 = __builtin_amdgcn_cvt_pk_u8_f32(red[2], 2u, d1); d1 = __builtin_amdgcn_cvt_pk_u8_f32(green[2], 3u, d1); d2 = __builtin_amdgcn_cvt_pk_u8_f32(blue[2], 0u, d2);
            d2 = __builtin_amdgcn_cvt_pk_u8_f32(red[3], 1u, d2); d2 = __builtin_amdgcn_cvt_pk_u8_f32(green[3], 2u, d2); d2 = __builtin_amdgcn_cvt_pk_u8_f32(blue[3], 3u, d2);
            const int y = py + k;
            typedef uint32_t Triple __attribute__((ext_vector_type(3), aligned(4)));
            __builtin_nontemporal_store(Triple{d0, d1, d2}, (Triple*)(out + (long)(a.top_down ? a.h - 1 - y : y)*pitch));
        }
    }
    }
}

// ---- bars / waveform as RUNS ------------------------------------------------------------------------------------------------------
// A bars or waveform frame is, column by column, a handful of runs of rows: every comparison of the fragments is `row < T` or
// `first <= row < first + count` with per-column integers (k_separable_axis), so along a column the colour changes at a few
// SPECIAL rows only — three per sample column for bars (where a height crosses the row pair of an output pixel), six for
// waveform — and between them red and green (all of waveform's channels) are constants. k_separable_fused evaluated the counts
// per pixel and row (64 VALU instructions per output pixel: a quarter of the chip's issue rate, 2.4 TB/s written); here one lane
// owns FOUR output pixels = 12 bytes = one `global_store_dwordx3`, walks SEP_RUN_ROWS rows with its eight column entries in
// registers and keeps the 12 bytes of the current run as a template:
//   fast row (no lane of the wave at a special row — 98-99 % of them): waveform stores the template; bars fills in blue, the one
//       channel that is not piecewise constant (bars.frag:17: the ramp 0.4·Σ·(1 − astuv.y)): per sample the RGBA8 byte of
//       `(below + ramp·(1 − y))·255` as fma(255·ramp, 1 − y, 255·below) → v_cvt_pk_u8 (2 instructions; the product is rounded once
//       instead of twice, which moves a byte only when the value sits within 10⁻⁵ of a half; the four bytes of a pixel are packed
//       in one register), then final.glsl's mean of the four bytes as (Σ + 2) >> 2 (v_sad_u8 with the 2 as its accumulator, a shift);
//   special row (wave-uniform branch on a ballot): every lane evaluates the row with the per-pixel counts of k_separable_fused and
//       rebuilds its template and its next special row.
// ≈ 13 instructions per output pixel for bars, 2 for waveform; a wave writes 768 contiguous bytes per row straight from registers
// (no LDS, no barrier). Exactness: red, green and waveform's channels are the generic chain's bytes (the resolve of 0/1 samples
// is exact, separable_fast.hpp above); blue's integer mean differs from the float chain of resolve_channel only when the four
// bytes sum to 2 (mod 4) — a tie that the float chain's rounding noise decides either way — by 1 LSB, the north star's bound.
constexpr int SEP_RUN_ROWS = 32;                                    // rows a wave walks; a block is 4 waves = 256 pixels x 128 rows

// bars: bytes of one pixel at sample row j0 from its two column entries (the per-pixel arithmetic of k_separable_fused)
__device__ __forceinline__ uint32_t count_byte(int count) { return (uint32_t)(count*255 + 2) >> 2; }               // RN(count*63.75), count 0..4
__device__ __forceinline__ uint32_t blue_mean(float b0, float b1, float b2, float b3, float ramp0, float ramp1, float omy0, float omy1) {
    uint32_t q = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ramp0, omy0, b0)*255.0f, 0u, 0u);
    q = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ramp1, omy0, b1)*255.0f, 1u, q);
    q = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ramp0, omy1, b2)*255.0f, 2u, q);
    q = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ramp1, omy1, b3)*255.0f, 3u, q);
    return __builtin_amdgcn_sad_u8(q, 0u, 2u) >> 2;
}
__device__ __forceinline__ uint32_t bars_pixel(const float4 c0, const float4 c1, int j0, float omy0, float omy1) {
    const int red = rows_of_pair(__float_as_int(c0.x), j0) + rows_of_pair(__float_as_int(c1.x), j0);
    const int green = rows_of_pair(__float_as_int(c0.y), j0) + rows_of_pair(__float_as_int(c1.y), j0);
    const int t0 = __float_as_int(c0.z), t1 = __float_as_int(c1.z);
    const uint32_t blue = blue_mean((j0 < t0) ? 1.0f : 0.0f, (j0 < t1) ? 1.0f : 0.0f, (j0 + 1 < t0) ? 1.0f : 0.0f, (j0 + 1 < t1) ? 1.0f : 0.0f,
                                    c0.w, c1.w, omy0, omy1);
    return count_byte(red) | (count_byte(green) << 8) | (blue << 16);
}
__device__ __forceinline__ uint32_t waveform_pixel(const float4 c0, const float4 c1, int j0, const uint8_t* lut) {
    return (uint32_t)lut[wave_pattern(__float_as_int(c0.x), __float_as_int(c1.x), j0)]
         | ((uint32_t)lut[wave_pattern(__float_as_int(c0.y), __float_as_int(c1.y), j0)] << 8)
         | ((uint32_t)lut[wave_pattern(__float_as_int(c0.z), __float_as_int(c1.z), j0)] << 16);
}
// the first pixel row >= `from` at which the row pair (2 py, 2 py + 1) meets the boundary `t` (a row count) of a column
__device__ __forceinline__ int special_after(int t, int from, int best) {
    const int s = t >> 1;
    return (s >= from && s < best) ? s : best;
}

// S == 2, w % 4 == 0. grid (ceil(w/256), ceil(h/(4*SEP_RUN_ROWS)), frames), block 256 = 4 waves, wave k walks rows [k*32, k*32 + 32).
// Only what a FAST row needs lives in registers across the walk (bars: per column the ramp and the blue state, both pre-multiplied by
// 255; the template; the next special row — ≈ 40 VGPRs, eight waves per SIMD); a special row reloads the lane's eight column entries
// (128 contiguous bytes, L2). The two per-row scalars of bars (1 − astuv.y of the pair's sample rows) are fetched ONCE per walk — lane
// l holds the value of sample row 2·y_first + l — and read with v_readlane: no memory latency inside the loop.
// (the kernel is stores: the more waves a SIMD holds, the more stores are in flight. Asked for, the compiler fits waveform's walk in
// 64 registers = 8 waves per SIMD instead of 70 = 7 — 283-293 us per 60 frames of 4K on three boxes against 294, inside their
// spread. bars' needs 91 = 5 waves; pressed into 80 or 64 it spills, and the spills are HBM writes of their own — 47 MB per
// launch at 6 waves, 308 -> 324 us — so it stays at 5)
template <int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KIND == SEP_BARS ? 5 : 8))) void k_separable_runs(const RenderArgs a, const SepTables t) {
    static_assert(KIND == SEP_BARS || KIND == SEP_WAVEFORM, "runs exist for the two comparison fragments");
    static_assert(SEP_RUN_ROWS == 32, "one lane per sample row of the walk");
    __shared__ uint8_t inside_lut[16];
    const int frame = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (KIND == SEP_WAVEFORM) {
        if (tid < 16) {
            uint32_t pattern[4];
#pragma unroll
            for (int k = 0; k < 4; k++) pattern[k] = unorm8(((tid >> k) & 1) ? 1.0f : 0.2f);
            inside_lut[tid] = (uint8_t)resolve_channel_any<2>(pattern, a.subsample, 0);
        }
        __syncthreads();
    }
    const int y_first = blockIdx.y*(4*SEP_RUN_ROWS) + wave*SEP_RUN_ROWS;
    if (y_first >= a.h) return;
    const int y_last = min(y_first + SEP_RUN_ROWS, a.h);
    const float4* rows = t.rows + (long)frame*a.hr;
    int omy_of_lane = 0;
    if constexpr (KIND == SEP_BARS) {                                 // before any lane leaves: v_readlane reads all of them
        const int sample_row = 2*y_first + lane;
        omy_of_lane = __float_as_int(rows[sample_row < (int)a.hr ? sample_row : (int)a.hr - 1].y);
    }
    const int px0 = (blockIdx.x*64 + lane)*4;                        // this lane's four output pixels
    const bool active = px0 < a.w;
    const float4* columns = t.columns + (long)frame*a.wr + (active ? 2*px0 : 0);
    uint8_t* out = (uint8_t*)a.out + (long)frame*a.out_frame_stride + (long)px0*3;
    const long pitch = (long)a.w*3;

    uint32_t d0 = 0, d1 = 0, d2 = 0;                                // the run's 12 bytes: R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
    float below255[8], ramp255[8];                                  // bars: 255 where the column's rows are below its blue height; 255 x its ramp
    int next = 0;
    auto rebuild = [&](int py) {                                    // state of the run that row `py` belongs to, and where it ends
        const int j0 = 2*py;
        uint32_t p[4];
        int best = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float4 c0 = columns[2*k], c1 = columns[2*k + 1];
            if constexpr (KIND == SEP_BARS) {
                const int red = rows_of_pair(__float_as_int(c0.x), j0) + rows_of_pair(__float_as_int(c1.x), j0);
                const int green = rows_of_pair(__float_as_int(c0.y), j0) + rows_of_pair(__float_as_int(c1.y), j0);
                p[k] = count_byte(red) | (count_byte(green) << 8);
                below255[2*k] = (j0 < __float_as_int(c0.z)) ? 255.0f : 0.0f;
                below255[2*k + 1] = (j0 < __float_as_int(c1.z)) ? 255.0f : 0.0f;
                ramp255[2*k] = c0.w*255.0f; ramp255[2*k + 1] = c1.w*255.0f;
                best = special_after(__float_as_int(c0.x), py, best); best = special_after(__float_as_int(c1.x), py, best);
                best = special_after(__float_as_int(c0.y), py, best); best = special_after(__float_as_int(c1.y), py, best);
                best = special_after(__float_as_int(c0.z), py, best); best = special_after(__float_as_int(c1.z), py, best);
            } else {
                p[k] = waveform_pixel(c0, c1, j0, inside_lut);
                const int runs[6] = {__float_as_int(c0.x), __float_as_int(c0.y), __float_as_int(c0.z), __float_as_int(c1.x), __float_as_int(c1.y), __float_as_int(c1.z)};
#pragma unroll
                for (int m = 0; m < 6; m++) {
                    const int first = runs[m] & 0xffff, count = (int)((unsigned)runs[m] >> 16);
                    best = special_after(first, py, best);
                    best = special_after(first + count, py, best);
                }
            }
        }
        d0 = p[0] | (p[1] << 24); d1 = (p[1] >> 8) | (p[2] << 16); d2 = (p[2] >> 16) | (p[3] << 8);
        next = best;
    };
    if (active) rebuild(y_first); else next = 0x7fffffff;
#pragma unroll 1
    for (int py = y_first; py < y_last; py++) {
        uint32_t o0, o1, o2;
        float omy0 = 0.0f, omy1 = 0.0f;
        if constexpr (KIND == SEP_BARS) {
            omy0 = __int_as_float(__builtin_amdgcn_readlane(omy_of_lane, 2*(py - y_first)));           // wave-uniform row index: SGPR lane selects
            omy1 = __int_as_float(__builtin_amdgcn_readlane(omy_of_lane, 2*(py - y_first) + 1));
        }
        if (__builtin_amdgcn_ballot_w64(py == next) != 0) {
            // a special row of some lane: the per-pixel evaluation for everybody, then the runs that start on the next row
            if (active) {
                uint32_t p[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float4 c0 = columns[2*k], c1 = columns[2*k + 1];
                    p[k] = (KIND == SEP_BARS) ? bars_pixel(c0, c1, 2*py, omy0, omy1) : waveform_pixel(c0, c1, 2*py, inside_lut);
                }
                o0 = p[0] | (p[1] << 24); o1 = (p[1] >> 8) | (p[2] << 16); o2 = (p[2] >> 16) | (p[3] << 8);
                rebuild(py + 1);
            }
        } else if constexpr (KIND == SEP_BARS) {
            const float y0 = omy0, y1 = omy1;                        // in VGPRs: a VALU operation with an SGPR operand issues at half rate
            uint32_t blue[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t q = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ramp255[2*k], y0, below255[2*k]), 0u, 0u);
                q = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ramp255[2*k + 1], y0, below255[2*k + 1]), 1u, q);
                q = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ramp255[2*k], y1, below255[2*k]), 2u, q);
                q = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(ramp255[2*k + 1], y1, below255[2*k + 1]), 3u, q);
                blue[k] = __builtin_amdgcn_sad_u8(q, 0u, 2u) >> 2;
            }
            o0 = d0 | (blue[0] << 16); o1 = d1 | (blue[1] << 8); o2 = d2 | blue[2] | (blue[3] << 24);
        } else {
            o0 = d0; o1 = d1; o2 = d2;
        }
        if (active) {
            uint32_t* row = (uint32_t*)(out + (long)(a.top_down ? a.h - 1 - py : py)*pitch);
            // non-temporal (`global_store_dwordx3 ... nt`): nobody on the device reads a frame back, so its lines need not stay in
            // the L2 — bars 308 -> 261 us per 60 frames of 4K (5.7 TB/s written = 0.71 of the HBM roof), waveform 293 -> 289 (its
            // time is the special rows around the wave, not the stores)
            typedef uint32_t Triple __attribute__((ext_vector_type(3), aligned(4)));
            __builtin_nontemporal_store(Triple{o0, o1, o2}, (Triple*)row);
        }
    }
}

// ---- default.glsl's singular line (round 6) ------------------------------------------------------------------------------------------
// hsv2rgb (shaderflow.glsl:406-425) takes `h = mod(h, TAU)` and switches on int(floor(6*(h/(2*PI)))) with cases 0..5 — and for a hue a hair
// below zero, `mod` returns TAU ITSELF in float32 (TAU + h rounds to TAU), the switch finds 6, no case applies and the colour is the
// function's initial grey instead of red. The set is a ray of the hue wheel less than 2.4e-7 rad wide: about ONE supersample per 4K frame falls
// into it by chance — and at iTau = 0 (the first frame of every export) the ray is the frame's diagonal x = y, whose samples sit on it
// EXACTLY: 71 pixels of that frame were up to 75 LSB from the reference next to the ring (found by a camera / hue sweep of
// tests/test_gpu_fullsize.py::test_basic_whole_frame_4k; the tiers above evaluate the wheel in its continuous form, which has no such line,
// and the generic kernels — the reference's own operations — have it). Fixed where it is cheap to be exact: this kernel walks the ray — per
// sample row and per sample column the three samples nearest to it, 36 000 threads per 4K frame —, evaluates the reference's chain for
// them, and where the switch falls through re-renders that pixel with the generic fragment (four samples, final.glsl's resolve) over what
// k_separable_fused wrote. < 1 % of the frame's time; bit-identical to the generic kernel on those pixels by construction.
__global__ __launch_bounds__(256) void k_default_singular(const RenderArgs a) {
    const int frame = blockIdx.y;
    const int k = blockIdx.x*256 + threadIdx.x;
    Uniforms u; Tex tex[TEX_HISTORY];
    frame_view(a, frame, u, tex);
    Frag f; f.u = &u; f.tex = tex; f.history = a.tex + TEX_HISTORY;
    // the ray: angle + (2*TAU*iTau) - PI/4 = 0 (mod TAU); iCamera.gluv is the fragment's gluv through an affine map per axis (identity or
    // axis-aligned camera: the only ones this path serves)
    // (once per block: float64 sine and cosine cost more than everything else a thread does here)
    __shared__ double ray[2];
    __shared__ float affine[4];
    __shared__ int usable;
    if (threadIdx.x == 0) {
        const double shift = 2.0*6.283185307179586*(double)u.iTau - 0.7853981633974483;
        const double theta = -shift - 6.283185307179586*::floor(-shift/6.283185307179586);
        ray[0] = ::cos(theta); ray[1] = ::sin(theta);
        bool behind = false;
        affine[0] = a.identity_camera ? 0.0f : camera_along_axis<0>(u, 0.0f, a.aspect, behind);
        affine[1] = a.identity_camera ? 0.0f : camera_along_axis<1>(u, 0.0f, a.aspect, behind);
        affine[2] = a.identity_camera ? 1.0f : camera_along_axis<0>(u, 1.0f, a.aspect, behind) - affine[0];
        affine[3] = a.identity_camera ? 1.0f : camera_along_axis<1>(u, 1.0f, a.aspect, behind) - affine[1];
        usable = (!behind && affine[2] != 0.0f && affine[3] != 0.0f) ? 1 : 0;
    }
    __syncthreads();
    if (!usable || k >= a.hr + a.wr) return;
    const double dx = ray[0], dy = ray[1];
    const float bx = affine[0], by = affine[1], mx = affine[2], my = affine[3];
    int ci, cj;                                                       // the candidate nearest to the ray on this thread's row / column
    if (k < a.hr) {
        cj = k;
        make_varyings(f, 0, cj, a.wr, a.hr, a.aspect, a.inv_wr, a.inv_hr);
        const double uy = (double)by + (double)my*(double)f.gluv.y;
        if (::fabs(dy) < 1e-9 || uy/dy <= 0.0) return;                // the ray does not cross this row
        const double gx = (uy*dx/dy - (double)bx)/(double)mx;        // gluv.x of the crossing
        const double at = ((gx/(double)a.aspect + 1.0)*0.5)*(double)a.wr - 0.5;
        if (!(at > -2.0 && at < (double)a.wr + 1.0)) return;
        ci = (int)::floor(at + 0.5);
    } else {
        ci = k - a.hr;
        make_varyings(f, ci, 0, a.wr, a.hr, a.aspect, a.inv_wr, a.inv_hr);
        const double ux = (double)bx + (double)mx*(double)f.gluv.x;
        if (::fabs(dx) < 1e-9 || ux/dx <= 0.0) return;
        const double gy = (ux*dy/dx - (double)by)/(double)my;
        const double at = ((gy + 1.0)*0.5)*(double)a.hr - 0.5;
        if (!(at > -2.0 && at < (double)a.hr + 1.0)) return;
        cj = (int)::floor(at + 0.5);
    }
    for (int d = -1; d <= 1; d++) {
        const int i = (k < a.hr) ? ci + d : ci, j = (k < a.hr) ? cj : cj + d;
        if (i < 0 || i >= a.wr || j < 0 || j >= a.hr) continue;
        make_varyings(f, i, j, a.wr, a.hr, a.aspect, a.inv_wr, a.inv_hr);
        const Camera cam = get_camera(f);
        if (cam.out_of_bounds) continue;
        // default.glsl:19-22 and hsv2rgb's first lines, the generic chain's operations
        const float angle = sf::atan2(cam.gluv.y, cam.gluv.x);
        const float h = sf::mod(angle + (2.0f*TAU*u.iTau) - (PI/4.0f), TAU);
        const int sector = sf::to_int(::floorf(6.0f*(h/(2.0f*PI))));
        if (sector >= 0 && sector <= 5) continue;
        // the pixel this sample belongs to, as the generic fused kernel renders it
        const int px = i >> 1, py = j >> 1;
        if (px >= a.w || py >= a.h) continue;
        uint32_t block[4];
        for (int q = 0; q < 4; q++) {
            make_varyings(f, min(2*px + (q & 1), a.wr - 1), min(2*py + (q >> 1), a.hr - 1), a.wr, a.hr, a.aspect, a.inv_wr, a.inv_hr);
            block[q] = pack_rgb8(rgb(shade<FRAG_DEFAULT>(f)));
        }
        uint8_t* out = (uint8_t*)a.out + (long)frame*a.out_frame_stride + ((long)(a.top_down ? a.h - 1 - py : py)*a.w + px)*3;
        out[0] = (uint8_t)resolve_channel_any<2>(block, a.subsample, 0);
        out[1] = (uint8_t)resolve_channel_any<2>(block, a.subsample, 8);
        out[2] = (uint8_t)resolve_channel_any<2>(block, a.subsample, 16);
    }
}

}  // namespace sf
