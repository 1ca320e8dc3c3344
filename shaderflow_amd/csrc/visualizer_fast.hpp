// visualizer_fast.hpp — visualizer.frag (examples/basic/shaders/visualizer.frag:6-74) under the freshly built camera, with
// everything that depends on the sample COLUMN only or on the sample ROW only evaluated once per frame instead of once per
// supersample (k_visualizer_axis → per-frame tables in HBM, L2-resident), and the fused fragment + resolve kernel reduced to
// what really is two-dimensional: the diagonal blur taps, the polar bar lookup and the final colour arithmetic.
//
// Under the identity camera (glsl.hpp camera_is_identity) iCamera.gluv == gluv, so for a W x H sample grid
//   * gluv.x, agluv.x, astuv.x, the background coordinate of the centre tap, the waveform tap texture(iWaveform, (astuv.x, 0)),
//     the vignette factor of x and `|gluv.x| > iWantAspect` take W distinct values per frame,
//   * gluv.y, …, `1 ∓ gluv.y` take H distinct values per frame,
// against W*H = 33.2 M supersamples at 4K 2xSSAA. The tables hold, per column (row):
//   q0 = { r, frac(r), byte offset of cell floor(r) inside the block's LDS tile, byte offset of the line's first cell }
//   q1 = { rot_c*u | -rot_s*u (col) or rot_s*u | rot_c*u (row), u*u, agluv*agluv }     (u = gluv.x or gluv.y)
//   q2 = { (sqrt(20)*astuv*(1-astuv))^vig_exp: this axis' factor of the frame's vignette, waveform.x*0.2 | 1-gluv.y, waveform.y*0.2 | 1+gluv.y, out of bounds }
//   q3..q6 = the AXIS LINE: the blur's taps along this axis (visualizer.frag:26-31, directions 0 and 180 degrees plus the
//            centre tap for a row of texels, 90 and 270 degrees for a column) summed per texel cell in closed form — for each
//            of the VIS_LINE_CELLS consecutive cells the line can touch: n = sum of tap weights, s = sum of weight*fraction.
//            A run of taps inside one cell sums to n*A + s*B' + (n*f)*C' + (s*f)*D (visualizer_kernels.hpp one_run), so the
//            20 + 1 taps of a line cost 8 cell fetches and 8*15 operations per supersample instead of 21 taps.
// `r` = the centre tap in texel units RELATIVE to the window of background cells the sample's block stages in LDS, so the
// tables are specific to the block geometry (BLOCK_COLUMNS x BLOCK_ROWS samples) they were built for.
//
// Bit-level contract: every term that decides a branch or a nearest-texel index — out of bounds, the polar angle and its bin,
// the bar height, both lengths compared with it, the waveform strips — is the generic chain's sequence of operations (same
// bits as fragments.hpp visualizer_post), merely evaluated where its inputs are known — or, for the polar chain, a cheaper
// evaluation whose error is bounded and which is re-run exactly by the whole wave whenever a lane comes within that bound of a
// decision (visualizer_fast_post). Colour-only terms use the hardware's log/exp/sqrt (1 ulp), as VisualizerShader does. The
// blur is VisualizerShader's (re-associated sums, <= 1e-6 relative).
#pragma once

#include "visualizer_kernels.hpp"

namespace sf {

constexpr int VIS_LINE_CELLS = 8;        // cells an axis line may touch: floor(2*radius) + 2 with radius <= 3.24 texels
constexpr int VIS_ENTRY_QUADS = 7;       // float4s per table entry

struct VisTables {
    float4* columns;                     // [frame][wr][VIS_ENTRY_QUADS]
    float4* rows;                        // [frame][hr][VIS_ENTRY_QUADS]
    int4* block_x;                       // [frame][blocks_x] = { x0, tw, fits, 0 }: window of background cells of a block column
    int4* block_y;                       // [frame][blocks_y] = { y0, th, fits, 0 }
    float4* ysteps;                      // [frame][hr][10] = { frac(y+), frac(y-), row bytes(y+), row bytes(y-) } of the diagonal taps' walk steps; may be null
    int blocks_x, blocks_y;              // blocks per row / column of the sample grid
    int block_columns, block_rows;       // samples per block along x / y
    int tile_pitch, tile_rows;           // LDS tile the offsets are computed for (cells)
    int cell_bytes;                      // distance between neighbouring cells of a tile row: 48 (float32 cells, three float4 together) or 8 (float16 cells in three planes)
    // the PIXEL tier (round 6, VisualizerStrip::pixel_tier): visualizer.frag:36-62's position-only factors once per output pixel
    float4* pixel_columns;               // [frame][w] = q1 of the pixel's centre column { rot_c*u, -rot_s*u, u*u, agluv.x^2 }; null: no pixel tier
    float4* pixel_rows;                  // [frame][h] = q1 of the pixel's centre row    { rot_s*u,  rot_c*u, u*u, agluv.y^2 }
    const float2* bars2;                 // [frame][spectrogram height][2 channels] = { bar amplitude sqrt(texel/1000), its largest difference to a neighbouring bin / the other channel at the ends }
    float pixel_reach_uv;                // how far a supersample can lie from its pixel's centre, in iCamera.gluv units (with a margin: 0.3 of the pixel's diagonal)
    float pixel_reach_agluv;             // ... in agluv units
    const uint8_t* wave_classes;         // [frame][blocks_y*blocks_x][8 waves]: 1 = the wave's tile takes the pixel tier (k_visualizer_classify)
};

// { amplitude, spread }: the bar of a spectrogram texel and how far the bars a NEIGHBOURING position could pick differ from it — the bins
// on either side (wrapped as the sampler wraps them) and, for the first and the last bin, the other channel (visualizer.frag:52 switches
// channel where music_uv.y changes sign, i.e. at circle = 0 and circle = 1). One thread per (frame, bin, channel).
__global__ void k_visualizer_bar_spread(const float* __restrict__ bars, long frame_stride, int frames, int height, int repeat_y, float2* __restrict__ out) {
    const long k = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= (long)frames*height*2) return;
    const int frame = (int)(k/(height*2)), e = (int)(k - (long)frame*height*2), bin = e >> 1, ch = e & 1;
    const float* b = bars + frame*frame_stride;
    const float mine = b[2*bin + ch];
    const float below = b[2*wrap_texel(bin - 1, height, repeat_y) + ch], above = b[2*wrap_texel(bin + 1, height, repeat_y) + ch];
    float spread = fmaxf(fabsf(below - mine), fabsf(above - mine));
    if (bin == 0 || bin == height - 1) spread = fmaxf(spread, fabsf(b[2*bin + (1 - ch)] - mine));
    out[k] = make_float2(mine, (spread == spread) ? spread : __builtin_inff());            // a NaN bar: never certain
}

// One axis of the sample grid: thread k builds the entry of column (AXIS == 0) or row (AXIS == 1) k of frame blockIdx.y.
template <int AXIS>
__device__ __forceinline__ void visualizer_axis_entry(const RenderArgs& a, const VisTables& t, int k, int frame) {
    const int n = AXIS == 0 ? a.wr : a.hr;
    if (k >= n) return;
    Uniforms u; Tex tex[TEX_HISTORY];
    frame_view(a, frame, u, tex);
    const VisualizerConsts c = a.vis_consts ? a.vis_consts[a.frame0 + frame] : (a.has_vis ? a.vis : visualizer_consts(u.iTime, u.iAudioVolume, u.iAudioSTD));
    const Tex& bg = tex[TEX_BACKGROUND];
    const float size = AXIS == 0 ? (float)bg.width : (float)bg.height;
    const float scale = AXIS == 0 ? a.bg_scale_x : 1.0f;              // gtexture's (height/width, 1), shaderflow.glsl:166-167
    const float aspect = AXIS == 0 ? a.aspect : 1.0f;                  // agluv2gluv, shaderflow.glsl:99
    const float offset = AXIS == 0 ? c.off_x : c.off_y;

    // vertex/default.glsl:1-17 and visualizer.frag:17-18 for ONE coordinate: the operations of make_varyings / visualizer_pre /
    // VisualizerShader::pre in the same order
    // `g` is the fragment's own gluv component (the strips of :72-73 and `out of bounds` use it), `uv` iCamera.gluv's: the same
    // under the identity camera, get_camera along this axis for a zoomed / panned one (glsl.hpp camera_is_axis_aligned)
    bool behind = false;
    auto centre_tap = [&](int index, float& ag, float& g, float& as, float& uv) {
        const float centre = ((float)index + 0.5f)/(float)n;
        ag = centre*2.0f - 1.0f;
        g = ag*aspect;
        as = (ag + 1.0f)/2.0f;
        uv = a.identity_camera ? g : camera_along_axis<AXIS>(u, g, a.aspect, behind);
        const float bgc = (((uv + 1.0f)/2.0f) - 0.5f)*c.zoom2 + 0.5f + offset;       // zoom(gluv2stuv(uv), z, 0.5) + offset
        const float st = ((((bgc*2.0f) - 1.0f)*scale) + 1.0f)/2.0f;                    // gluv2stuv(stuv2gluv(bg)*scale)
        return st*size - 0.5f;
    };
    float ag, g, as, uv;
    const float centre = centre_tap(k, ag, g, as, uv);

    // the window of the block this column/row belongs to: VisualizerShader::setup 1a (two corner samples bound all of them)
    const int per_block = AXIS == 0 ? t.block_columns : t.block_rows;
    const int block = k / per_block;
    const int first_index = block*per_block, last_index = min(first_index + per_block, n) - 1;
    float dummy0, dummy1, dummy2, dummy3;
    const float c_first = centre_tap(first_index, dummy0, dummy1, dummy2, dummy3), c_last = centre_tap(last_index, dummy0, dummy1, dummy2, dummy3);
    const float lo = fminf(c_first, c_last), hi = fmaxf(c_first, c_last);
    const float texels = fabsf(c.intensity*scale*size);               // blur radius in texels (glsl.hpp gtexture)
    const float reach = texels*1.101f + 0.001f;
    const int limit = AXIS == 0 ? t.tile_pitch : t.tile_rows;
    int origin = 0, extent = 0, fits = 0;
    if ((c.intensity == c.intensity) && fabsf(lo) < 1e8f && fabsf(hi) < 1e8f && reach < 64.0f) {
        origin = (int)floorf(lo - reach);
        extent = (int)floorf(hi + reach) - origin + 1;
        // the line's cells must exist in the staged window even when the music is silent
        if (extent < VIS_LINE_CELLS) extent = VIS_LINE_CELLS;
        fits = (extent <= limit) ? 1 : 0;
    }
    // the axis line: weights per texel cell, in tile-local coordinates
    const float r = centre - (float)origin;
    const float ax = c.intensity*a.bg_scale_x*(float)bg.width;        // texels per unit displacement, as VisualizerShader::blur_tile uses it for both axes
    const float first = a.tap_x[0]*ax, step = (a.tap_x[1] - a.tap_x[0])*ax;
    const float reach_line = fabsf(first) + 9.0f*fabsf(step);
    // a line touches at most floor(2*reach) + 2 consecutive cells: a property of the frame, so every column and row of a block
    // agrees on it and the block's flag needs no voting
    if (!(2.0f*reach_line + 1.0e-3f < (float)(VIS_LINE_CELLS - 1))) fits = 0;
    if (k == first_index) (AXIS == 0 ? t.block_x : t.block_y)[(long)frame*(AXIS == 0 ? t.blocks_x : t.blocks_y) + block] = make_int4(origin, extent, fits, 0);
    const float cell_r = floorf(r), frac_r = r - cell_r;
    // slot 0 of the line: the leftmost cell it can touch, pulled back so that all VIS_LINE_CELLS slots lie inside the window
    int start = (int)floorf(r - reach_line);
    start = max(0, min(start, extent - VIS_LINE_CELLS));
    float wn[VIS_LINE_CELLS], ws[VIS_LINE_CELLS];
#pragma unroll
    for (int s = 0; s < VIS_LINE_CELLS; s++) { wn[s] = 0.0f; ws[s] = 0.0f; }
    auto tap = [&](float position, float weight) {
        const float cell = floorf(position);
        const float fraction = position - cell;
        const int slot = (int)cell - start;                           // 0 <= slot < VIS_LINE_CELLS whenever `fits`
#pragma unroll
        for (int s = 0; s < VIS_LINE_CELLS; s++)
            if (s == slot) { wn[s] = wn[s] + weight; ws[s] = fmaf(weight, fraction, ws[s]); }
    };
    if (AXIS == 0) tap(r, 1.0f);                                      // the centre tap (visualizer.frag:19) rides on the row of texels
    for (int j = 0; j < 10; j++) {
        const float d = fmaf((float)j, step, first);
        tap(r + d, AXIS == 0 ? 2.0f : 1.0f);                          // direction 0 is direction 8 as well (float loop counter, :26)
        tap(r - d, 1.0f);
    }

    float4* e = (AXIS == 0 ? t.columns : t.rows) + ((long)frame*n + k)*VIS_ENTRY_QUADS;
    // this axis' factor of the vignette: pow(vig.x*vig.y*20, e) = (sqrt(20)*ax(1-ax))^e * (sqrt(20)*ay(1-ay))^e with the frame's
    // exponent (visualizer.frag:65-66) — one exp2 per table entry instead of one per supersample
    const float vignette = __builtin_amdgcn_exp2f(c.vig_exp*(__builtin_amdgcn_logf(as*(1.0f - as)) + 2.1609640475f));
    const int cell_bytes = AXIS == 0 ? t.cell_bytes : t.tile_pitch*t.cell_bytes;
    e[0] = make_float4(r, frac_r, __int_as_float((int)cell_r*cell_bytes), __int_as_float(start*cell_bytes));
    if (AXIS == 0) {
        // rotate2d(-PI/2)*uv (visualizer.frag:39): x' = rot_c*x + rot_s*y, y' = (-rot_s)*x + rot_c*y — the products of x
        e[1] = make_float4(c.rot_c*uv, (-c.rot_s)*uv, uv*uv, ag*ag);
        const Tex& wave = tex[TEX_WAVEFORM];
        const vec2 w = texture_xy(wave, vec2{as, 0.0f});                                       // :71
        e[2] = make_float4(vignette, 0.2f*w.x, 0.2f*w.y, __int_as_float((behind || sf::abs(g) > u.iWantAspect) ? 1 : 0));   // camera.glsl:83
    } else {
        e[1] = make_float4(c.rot_s*uv, c.rot_c*uv, uv*uv, ag*ag);
        e[2] = make_float4(vignette, 1.0f - g, 1.0f + g, 0.0f);   // :72-73 compare these with the waveform
        if (t.ysteps) {
            // The y half of a diagonal tap — fraction and row of cells of y +- k*s (VisualizerShader::blur_tile's walk) — depends
            // on the sample ROW and the walk step only: 10 entries per row instead of eight instructions per tap quadruple of
            // every sample, and wave-uniform for a wave whose lanes share their rows
            const float dstep = (a.tap_x[11] - a.tap_x[10])*ax, dfirst = a.tap_x[10]*ax;
            const float row_bytes = (float)(t.tile_pitch*t.cell_bytes);
            float4* ys = t.ysteps + ((long)frame*n + k)*10;
            for (int w = 0; w < 10; w++) {
                const float d = fmaf((float)w, dstep, dfirst);
                const float yp = r + d, ym = r - d;
                const float ayp = __builtin_amdgcn_fractf(yp), aym = __builtin_amdgcn_fractf(ym);
                ys[w] = make_float4(ayp, aym, __int_as_float((int)((yp - ayp)*row_bytes)), __int_as_float((int)((ym - aym)*row_bytes)));
            }
        }
    }
    e[3] = make_float4(wn[0], ws[0], wn[1], ws[1]);
    e[4] = make_float4(wn[2], ws[2], wn[3], ws[3]);
    e[5] = make_float4(wn[4], ws[4], wn[5], ws[5]);
    e[6] = make_float4(wn[6], ws[6], wn[7], ws[7]);
    // the pixel tier's entry of output pixel k along this axis: the same terms at the pixel's CENTRE (between its supersamples)
    const int pixels = AXIS == 0 ? a.w : a.h;
    if (t.pixel_columns && k < pixels) {
        const float pc = ((float)k + 0.5f)/(float)pixels;
        const float pag = pc*2.0f - 1.0f, pg = pag*aspect;
        bool pbehind = false;
        const float puv = a.identity_camera ? pg : camera_along_axis<AXIS>(u, pg, a.aspect, pbehind);
        (AXIS == 0 ? t.pixel_columns : t.pixel_rows)[(long)frame*pixels + k] =
            AXIS == 0 ? make_float4(c.rot_c*puv, (-c.rot_s)*puv, puv*puv, pag*pag) : make_float4(c.rot_s*puv, c.rot_c*puv, puv*puv, pag*pag);
    }
}
// both axes in ONE launch (the two tables are latency-bound chains of a few thousand threads each: side by side they take the time
// of one — 53 instead of 105 us per 60 frames of 4K): blocks [0, ceil(wr/128)) build columns, the rest rows; frame = blockIdx.y
__global__ __launch_bounds__(128) void k_visualizer_axes(const RenderArgs a, const VisTables t) {
    const int column_blocks = (a.wr + 127)/128;
    if ((int)blockIdx.x < column_blocks) visualizer_axis_entry<0>(a, t, blockIdx.x*128 + threadIdx.x, blockIdx.y);
    else visualizer_axis_entry<1>(a, t, ((int)blockIdx.x - column_blocks)*128 + threadIdx.x, blockIdx.y);
}

// visualizer.frag:32-73 (fragments.hpp visualizer_post<true>) on the blur's sums and the separable terms of the tables: c1/c2 the
// column's q1/q2, r1/r2 the row's. Returns the RGB8 texel iScreen would hold (alpha is never read by final.glsl).
//
// The polar part — atan(y, x)/PI -> bin of the spectrogram -> bar height -> two comparisons with length(uv) — decides branches
// and a nearest-texel index, so it has to be the generic chain's bits: three IEEE divisions and an IEEE square root, 55 of this
// function's ≈ 190 instructions. They are SPECULATED: the same formulas with v_rcp_f32/v_sqrt_f32 (1 ulp) first, which differ from
// the exact chain by <= 1.8e-7 in `circle` (tools/check_polar_speculation.py, 4 M points with the reciprocals perturbed by an ulp
// either way) and by an ulp in `len`; a lane is CERTAIN when circle*height stays height*1e-6 + 2e-6 away from an integer (its bin
// cannot differ) and len stays 5e-4 away from both radii (no comparison can differ, and pow(len - rr, 0.05), steep near zero,
// moves by < 0.1 LSB). If any lane of the wave is not, the whole wave evaluates the exact chain as well (1-3 % of the row-waves:
// those that a bin boundary or the ring crosses within the margin) — wave-uniform, so no lane ever mixes the two.
#ifndef VIS_SPECULATE
#define VIS_SPECULATE 1
#endif

template <bool WITH_ALPHA = false>                               // the fused kernels never look at alpha (final.glsl takes .rgb); iScreen holds it
__device__ __forceinline__ uint32_t visualizer_fast_post(const RenderArgs& a, int frame, const VisualizerConsts& c, float r, float g, float b,
                                                         const float4 c1, const float4 c2, const float4 r1, const float4 r2) {
    const float norm = 1.0f/(255.0f*10.0f*8.0f);                      // (sum/255)/(quality*directions), visualizer.frag:32
    const vec3 space = vec3{1.0f, 11.0f, 26.0f}/255.0f;                                                // :9
    vec3 col;
    {
        const float la = __builtin_amdgcn_sqrtf(c1.w + r1.w);                                   // length(agluv), colour only
        const float cl = clamp01(la - 0.3f);
        const float c2l = cl*cl;
        const float flash = norm*(1.0f + c.flash*(c2l*c2l*c2l));                                // :32 and :36 in one factor
        col = vec3{r*flash, g*flash, b*flash};
    }
    const vec2 music_uv = vec2{c1.x + r1.x, c1.y + r1.y}*c.shrink;                              // :39-40
    const float radius = 0.17f;
    const Tex& sp = a.tex[TEX_SPECTROGRAM];
    const float* bars = a.tape_bars + (long)(a.frame0 + frame)*a.spectrogram_stride;
    const float height = (float)sp.height;
    const float square = music_uv.x*music_uv.x + music_uv.y*music_uv.y;
    // length(iCamera.gluv): :62 wants it /20 for the colour, and music_uv is that vector rotated and shrunk (:39-40), so its length —
    // for the lanes that are certain of their side of both radii by 5e-4 — is this one times |shrink| (one square root for two)
    const float reach = __builtin_amdgcn_sqrtf(c1.z + r1.z);

    float len, bar, rr;
    auto heights = [&](float circle_, float len_) {                                             // :45-46, :52, given the angle
        const int bin = wrap_texel((int)::floorf(circle_*height), sp.height, sp.repeat_y);
        const float amplitude = bars[2*bin + ((music_uv.y < 0.0f) ? 0 : 1)];                    // sqrt(texel/1000), the channel :52 picks
        bar = amplitude*(0.05f + 3.0f*smoothstep01(circle_*0.5f));                              // smoothstep(0, 2, circle): x/2 is exact
        rr = radius + 0.5f*bar;
        len = len_;
    };
    bool certain = false;
    if (VIS_SPECULATE) {
        const float ax = sf::abs(music_uv.x), ay = sf::abs(music_uv.y);
        const float hi = __builtin_fmaxf(ax, ay), lo = __builtin_fminf(ax, ay);
        const float t = lo*__builtin_amdgcn_rcpf(hi);                                           // hi == 0: NaN, never certain
        const bool upper = t > 0x1.a8279ap-2f;
        const float u = upper ? (t - 1.0f)*__builtin_amdgcn_rcpf(t + 1.0f) : t;
        const float z = u*u;
        float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
        p = fmaf(p, z, 1.99777106478e-1f);
        p = fmaf(p, z, -3.33329491539e-1f);
        float angle = (upper ? QUARTER_PI : 0.0f) + fmaf(p*z, u, u);
        if (ay > ax) angle = HALF_PI - angle;
        if (music_uv.x < 0.0f) angle = 0x1.921fb6p+1f - angle;                                  // |atan(y, x)|: the sign of y does not matter
        const float approximate = angle*0x1.45f306p-2f;                                         // /PI
        heights(approximate, reach*sf::abs(c.shrink));
        const float scaled = approximate*height;
        const float fraction = scaled - ::floorf(scaled);
        const float margin = fmaf(height, 1.0e-6f, 2.0e-6f);
        certain = (fraction > margin) && (fraction < 1.0f - margin) && (sf::abs(len - radius) > 5.0e-4f) && (sf::abs(len - rr) > 5.0e-4f);
    }
    if (!VIS_SPECULATE || __builtin_amdgcn_ballot_w64(!certain) != 0)
        heights(sf::abs(atan1n(music_uv)), sf::sqrt(square));                                   // :44 and length(music_uv), the generic chain's bits

    if (len < radius) {                                                                         // :49-50
        col = col*0.5f;
    } else {
        if (len < rr) col = mix(col, vec3{1.0f, 1.0f, 1.0f}, smoothstep01(0.5f + bar));         // :56
        else col = col*ColourMath<true>::pow((len - rr)*0.5f, 0.05f);                           // :58
    }
    {
        const float lp = reach*0.05f;                                                            // length(uv)/20, colour only
        col = mix(col, space, smoothstep01(lp));                                                // :62
    }
    // the vignette (:65-66) = column factor * row factor (k_visualizer_axis); the unorm8 scale rides on it
    col = col*((c2.x*r2.x)*255.0f);
    // the waveform strips along the top and bottom edges and the bars outside the wanted aspect: most waves have no such lane
    const bool strip_top = r2.y < c2.y, strip_bottom = r2.z < c2.z, outside = __float_as_int(c2.w) != 0;
    uint32_t alpha = 0xff000000u;                                                               // fragColor.a = 1 (:68)
    if (__builtin_amdgcn_ballot_w64(strip_top || strip_bottom || outside) != 0) {
        float opacity = 1.0f;
        if (strip_top) { col = col*0.8f; opacity = opacity*0.8f; }                              // :72 scales the whole vec4
        if (strip_bottom) { col = col*0.8f; opacity = opacity*0.8f; }                           // :73
        if (outside) { col = space*255.0f; opacity = 0.0f; }                                    // :11-14
        if (WITH_ALPHA) alpha = unorm8(opacity) << 24;
    }
    uint32_t texel = __builtin_amdgcn_cvt_pk_u8_f32(col.x, 0u, 0u);                            // pack_rgb8 of col/255
    texel = __builtin_amdgcn_cvt_pk_u8_f32(col.y, 1u, texel);
    texel = __builtin_amdgcn_cvt_pk_u8_f32(col.z, 2u, texel);
    return WITH_ALPHA ? (texel | alpha) : texel;
}

// ---- the pixel tier (round 6) ------------------------------------------------------------------------------------------------------
// visualizer.frag:36-62 is, per channel, AFFINE in the blur's sum: texel_c = (sum_c + space_c*w)*A*vignette with
//     A = norm*flash*ring*(1 - sm)*255,  w = sm/(norm*flash*ring*(1 - sm))      (flash :36, ring = 0.5 | pow((len - rr)/2, 0.05) :49-58, sm :62)
// and A, w depend on the sample's POSITION only — through the most expensive part of the fragment (the polar chain: 70 of visualizer_fast_post's
// 110 instructions). The supersamples of a pixel lie a quarter of a pixel from its centre, so where A varies by less than a fraction of an LSB
// across the pixel one evaluation at the centre serves them all: the strip kernel then spends eleven instructions per supersample (vignette,
// strips, three multiply-adds, three products, three conversions) instead of 110, and the position-only part once per pixel ROW of a lane's column.
//
// WHERE that holds is decided per WAVE TILE (the 64 sample columns x WALK sample rows of one wave) by k_visualizer_classify, one thread per
// tile and frame, BEFORE the strip kernel runs — so a wave knows from one byte which of its two paths it takes, nothing of the other path is
// alive in its registers, and no lane ever mixes the two. The tile's rectangle of whole output pixels is bounded in music_uv space (a rotation
// and a uniform scale of iCamera.gluv: still a rectangle): len_min / len_max = the distances of its nearest / farthest point from the origin,
// the bins its angles can reach (+ one on either side), the largest bar and the largest bar-to-neighbour difference among them
// (k_visualizer_bar_spread). With `reach` = the largest distance of a supersample from its pixel's centre (VisTables::pixel_reach_*):
//   * the ring's gain: all inside the disc (len_max < radius - reach): the constant 0.5; all outside every bar the tile can see
//     (len_min - reach > rr_hi + drr_hi): pow(x, 0.05), x = (len - rr)/2, whose relative change over |dx| <= (reach + drr_hi)/2 is at most
//     0.05*dx/(x_min - dx); anything else (the disc's edge, the bars, their outline) is the per-sample path's;
//   * the flash 1 + f*cl^6 changes by at most 6*f*cl_max^5*reach_agluv/(1 + f*cl_min^6) of itself;
//   * sm = smoothstep(length(uv)/20) moves by < 1e-5 of itself over a pixel: nothing.
// A tile takes the pixel tier when (ring + flash bounds)*255 < 0.4 LSB: every supersample's texel then differs from its own evaluation by less
// than half an LSB BEFORE quantisation, so after final.glsl's mean a frame differs from the per-sample kernel's by at most 1 LSB (in practice
// by one in a few per cent of the values, none further: tests/test_gpu_fullsize.py compares both with the oracle).
struct PixelGains { float A, w; };          // texel_c = (sum_c + space_c*w)*A*vignette, space = (1, 11, 26) (:9; A carries the 255, w the sm/(1 - sm))
__device__ __forceinline__ PixelGains visualizer_pixel_gains(const RenderArgs& a, const VisTables& t, int frame, const VisualizerConsts& c, const float4 pc1, const float4 pr1) {
    const float norm = 1.0f/(255.0f*10.0f*8.0f);
    const float radius = 0.17f;
    PixelGains g;
    // :36 the flash
    const float la = __builtin_amdgcn_sqrtf(pc1.w + pr1.w);
    const float cl = clamp01(la - 0.3f);
    const float c2l = cl*cl;
    const float fl = fmaf(c.flash, c2l*c2l*c2l, 1.0f);
    // :39-46 the polar chain at the centre (visualizer_fast_post's speculated form: the centre needs no exact bits — a tile whose bars could
    // change the gain by a fraction of an LSB is not here)
    const float mx = (pc1.x + pr1.x)*c.shrink, my = (pc1.y + pr1.y)*c.shrink;
    const float reach = __builtin_amdgcn_sqrtf(pc1.z + pr1.z);
    const float len = reach*sf::abs(c.shrink);
    const float ax = sf::abs(mx), ay = sf::abs(my);
    const float hi = __builtin_fmaxf(ax, ay), lo = __builtin_fminf(ax, ay);
    const float q = lo*__builtin_amdgcn_rcpf(hi);
    const bool upper = q > 0x1.a8279ap-2f;
    const float u = upper ? (q - 1.0f)*__builtin_amdgcn_rcpf(q + 1.0f) : q;
    const float z = u*u;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    float angle = (upper ? QUARTER_PI : 0.0f) + fmaf(p*z, u, u);
    if (ay > ax) angle = HALF_PI - angle;
    if (mx < 0.0f) angle = 0x1.921fb6p+1f - angle;
    const float circle = angle*0x1.45f306p-2f;
    const Tex& sp = a.tex[TEX_SPECTROGRAM];
    const int bin = wrap_texel((int)::floorf(circle*(float)sp.height), sp.height, sp.repeat_y);
    const float amplitude = t.bars2[(long)frame*sp.height*2 + 2*bin + ((my < 0.0f) ? 0 : 1)].x;
    const float rr = fmaf(0.5f*(0.05f + 3.0f*smoothstep01(circle*0.5f)), amplitude, radius);
    const float ring = (len < radius) ? 0.5f : ColourMath<true>::pow((len - rr)*0.5f, 0.05f);          // (a classified tile is all inside or all outside the bars)
    const float sm = smoothstep01(reach*0.05f);                                                         // :62
    // mix(col, space, sm)*255 = (sum*(norm*flash*ring)*(1 - sm) + space*sm)*255 = (sum + space*w)*A
    g.A = ((norm*255.0f)*fl)*(ring*(1.0f - sm));
    g.w = sm*__builtin_amdgcn_rcpf(g.A);                                                                // (space_c = {1, 11, 26}/255: the 255 cancels against A's)
    return g;
}

// One thread per (frame, wave tile): 1 = the tile takes the pixel tier. Launched after k_visualizer_axes / k_visualizer_bar_spread.
template <int S, int WALK, int COLUMN_GROUPS>
__global__ __launch_bounds__(256) void k_visualizer_classify(const RenderArgs a, const VisTables t, uint8_t* __restrict__ classes) {
    constexpr int ROW_GROUPS = 8/COLUMN_GROUPS, COLS = 64*COLUMN_GROUPS, RROWS = ROW_GROUPS*WALK;
    const long tiles = (long)t.blocks_x*t.blocks_y*8;
    const long k = (long)blockIdx.x*blockDim.x + threadIdx.x;
    const int frame = blockIdx.y;
    if (k >= tiles) return;
    const int wave = (int)(k & 7), block = (int)(k >> 3), bx = block % t.blocks_x, by = block / t.blocks_x;
    const int group = wave % ROW_GROUPS, x0 = bx*COLS + (wave/ROW_GROUPS)*64, y0 = by*RROWS + group*WALK;
    uint8_t verdict = 0;
    Uniforms u; Tex tex[TEX_HISTORY];
    frame_view(a, frame, u, tex);
    const VisualizerConsts c = a.vis_consts ? a.vis_consts[a.frame0 + frame] : (a.has_vis ? a.vis : visualizer_consts(u.iTime, u.iAudioVolume, u.iAudioSTD));
    // the tile's whole output pixels: columns [px0, px1], rows [py0, py1] (a tile beyond the frame's edge has no samples: no verdict needed)
    const int px0 = x0/S, px1 = min(x0 + 63, a.wr - 1)/S, py0 = y0/S, py1 = min(y0 + WALK - 1, a.hr - 1)/S;
    if (x0 < a.wr && y0 < a.hr && c.flash >= 0.0f && c.shrink == c.shrink) {
        // the rectangle's edges in iCamera.gluv and agluv (the camera is the identity or acts per axis: glsl.hpp camera_along_axis)
        bool behind = false;
        auto edge_x = [&](int pixel_edge, float& ag) { ag = (float)pixel_edge/(float)a.w*2.0f - 1.0f; const float g = ag*a.aspect; return a.identity_camera ? g : camera_along_axis<0>(u, g, a.aspect, behind); };
        auto edge_y = [&](int pixel_edge, float& ag) { ag = (float)pixel_edge/(float)a.h*2.0f - 1.0f; return a.identity_camera ? ag : camera_along_axis<1>(u, ag, a.aspect, behind); };
        float agx0, agx1, agy0, agy1;
        const float ux0 = edge_x(px0, agx0), ux1 = edge_x(px1 + 1, agx1), uy0 = edge_y(py0, agy0), uy1 = edge_y(py1 + 1, agy1);
        // music_uv = rotate2d(-PI/2)*uv*shrink (:39-40): the corners, through the same products the tables hold
        float mx[4], my[4];
        const float cx[4] = {ux0, ux1, ux0, ux1}, cy[4] = {uy0, uy0, uy1, uy1};
        for (int q = 0; q < 4; q++) { mx[q] = (c.rot_c*cx[q] + c.rot_s*cy[q])*c.shrink; my[q] = ((-c.rot_s)*cx[q] + c.rot_c*cy[q])*c.shrink; }
        const float shrink = sf::abs(c.shrink);
        // distances: |uv| over the rectangle [ulo, uhi] x [vlo, vhi]
        const float ulo = fminf(ux0, ux1), uhi = fmaxf(ux0, ux1), vlo = fminf(uy0, uy1), vhi = fmaxf(uy0, uy1);
        const float nx = fmaxf(fmaxf(ulo, -uhi), 0.0f), ny = fmaxf(fmaxf(vlo, -vhi), 0.0f);              // the nearest point's |u|, |v|
        const float fx = fmaxf(fabsf(ulo), fabsf(uhi)), fy = fmaxf(fabsf(vlo), fabsf(vhi));              // the farthest corner's
        const float len_min = sqrtf(nx*nx + ny*ny)*shrink*(1.0f - 1.0e-5f), len_max = sqrtf(fx*fx + fy*fy)*shrink*(1.0f + 1.0e-5f);
        const float radius = 0.17f, far = t.pixel_reach_uv*shrink;
        // the flash: cl over the rectangle in agluv
        const float anx = fmaxf(fmaxf(fminf(agx0, agx1), -fmaxf(agx0, agx1)), 0.0f), any = fmaxf(fmaxf(fminf(agy0, agy1), -fmaxf(agy0, agy1)), 0.0f);
        const float afx = fmaxf(fabsf(agx0), fabsf(agx1)), afy = fmaxf(fabsf(agy0), fabsf(agy1));
        const float cl_min = fminf(fmaxf(sqrtf(anx*anx + any*any) - 0.3f, 0.0f), 1.0f), cl_max = fminf(fmaxf(sqrtf(afx*afx + afy*afy) - 0.3f, 0.0f), 1.0f);
        const float cl_min2 = cl_min*cl_min, cl_max2 = cl_max*cl_max;
        const float rel_flash = c.flash*6.0f*t.pixel_reach_agluv*(cl_max2*cl_max2*cl_max)/(1.0f + c.flash*cl_min2*cl_min2*cl_min2);
        float rel_ring = __builtin_inff();
        if (len_max < radius - far) rel_ring = 0.0f;                                                    // all of it inside the disc (:49-50)
        else if (nx > 0.0f || ny > 0.0f) {
            // the angles the tile can see: `circle` = |atan(y, x)|/PI over a convex region that does not hold the origin takes its extremes at
            // corners — unless the region crosses the line y = 0, where circle reaches 0 (x > 0) or 1 (x < 0) and the channel switches
            const Tex& sp = tex[TEX_SPECTROGRAM];
            float lo = 2.0f, hi = -1.0f, ylo = my[0], yhi = my[0], xlo = mx[0], xhi = mx[0];
            for (int q = 0; q < 4; q++) {
                const float circle = sf::abs(atan1n(vec2{mx[q], my[q]}));
                lo = fminf(lo, circle); hi = fmaxf(hi, circle);
                ylo = fminf(ylo, my[q]); yhi = fmaxf(yhi, my[q]); xlo = fminf(xlo, mx[q]); xhi = fmaxf(xhi, mx[q]);
            }
            const bool crosses = ylo <= 0.0f && yhi >= 0.0f;
            if (crosses) { if (xhi > 0.0f) lo = 0.0f; if (xlo < 0.0f) hi = 1.0f; }
            // (the rectangle holds every supersample and every pixel centre of the tile; a hundredth of a bin of slack for the corners' own
            // rounding and the centres' speculated angles, 2e-5 bins from the exact ones. What a pixel AT a bin's edge sees of the neighbouring
            // bin is that bin's `spread`, which k_visualizer_bar_spread took over both neighbours)
            const int b0 = max((int)floorf(lo*(float)sp.height - 0.01f), 0), b1 = min((int)floorf(hi*(float)sp.height + 0.01f), sp.height - 1);
            const float2* bars = t.bars2 + (long)frame*sp.height*2;
            float amp = 0.0f, spread = 0.0f;
            bool finite = true;
            for (int b = b0; b <= b1; b++)
                for (int ch = (crosses || ylo < 0.0f) ? 0 : 1; ch <= ((crosses || yhi >= 0.0f) ? 1 : 0); ch++) {
                    const float2 e = bars[2*b + ch];
                    amp = fmaxf(amp, e.x); spread = fmaxf(spread, e.y);
                    finite = finite && (e.x == e.x) && (e.y < __builtin_inff());
                }
            const float t_hi = fminf(fmaxf(hi*0.5f, 0.0f), 1.0f), F_hi = 0.05f + 3.0f*(t_hi*t_hi*(3.0f - 2.0f*t_hi));
            // F itself moves by at most 2.25 per unit of circle: over a pixel's angular half-width reach/len, in units of circle reach/(PI*len)
            const float rr_hi = radius + 0.5f*F_hi*amp, drr = 0.5f*F_hi*spread + 1.2f*(t.pixel_reach_uv/(PI*fmaxf(len_min/shrink, 1.0e-6f)))*amp;
            const float x_min = (len_min - rr_hi)*0.5f, dx = (far + drr)*0.5f + 1.0e-6f;
            if (finite && x_min - dx > 0.0f) rel_ring = 0.05f*dx/(x_min - dx);
        }
        if (!behind && (rel_ring + rel_flash)*255.0f < 0.4f) verdict = 1;                               // (a NaN anywhere: not the tier)
    }
    classes[(long)frame*tiles + k] = verdict;
}

// Stages the window [x0, x0+tw) x [y0, y0+th) of the background as difference-basis cells (VisualizerShader::setup step 2)
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
// PLANE_CELLS == 0: float32 cells, a cell's three float4 side by side (48 bytes); otherwise float16 cells — every component is an
// integer of magnitude <= 510, exact in float16 — as three planes of one half4 (8 bytes) per cell, PLANE_CELLS cells apart
template <int TILE_PITCH, int THREADS, int PLANE_CELLS = 0>
__device__ __forceinline__ void visualizer_fast_stage(const Tex& bg, float4* cells, int x0, int y0, int tw, int th, int tid) {
    const uint8_t* data = (const uint8_t*)bg.data;
    const int comps = bg.components;
    typedef uint32_t unaligned_u32 __attribute__((aligned(1)));
    for (int idx = tid; idx < TILE_PITCH*th; idx += THREADS) {
        const int ty = idx / TILE_PITCH, tx = idx - ty*TILE_PITCH;
        if (tx >= tw) continue;
        const int j0 = wrap_texel(y0 + ty, bg.height, bg.repeat_y), j1 = wrap_texel(y0 + ty + 1, bg.height, bg.repeat_y);
        const int i0 = wrap_texel(x0 + tx, bg.width, bg.repeat_x), i1 = wrap_texel(x0 + tx + 1, bg.width, bg.repeat_x);
        const uint32_t row0 = (uint32_t)j0*(uint32_t)bg.width, row1 = (uint32_t)j1*(uint32_t)bg.width;
        const uint32_t w00 = *(const unaligned_u32*)(data + (size_t)(row0 + i0)*comps);
        const uint32_t w10 = *(const unaligned_u32*)(data + (size_t)(row0 + i1)*comps);
        const uint32_t w01 = *(const unaligned_u32*)(data + (size_t)(row1 + i0)*comps);
        const uint32_t w11 = *(const unaligned_u32*)(data + (size_t)(row1 + i1)*comps);
        const float r00 = (float)(w00 & 255u), g00 = (float)((w00 >> 8) & 255u), b00 = (float)((w00 >> 16) & 255u);
        const float r10 = (float)(w10 & 255u), g10 = (float)((w10 >> 8) & 255u), b10 = (float)((w10 >> 16) & 255u);
        const float r01 = (float)(w01 & 255u), g01 = (float)((w01 >> 8) & 255u), b01 = (float)((w01 >> 16) & 255u);
        const float r11 = (float)(w11 & 255u), g11 = (float)((w11 >> 8) & 255u), b11 = (float)((w11 >> 16) & 255u);
        if constexpr (PLANE_CELLS != 0) {
            half4* cell = (half4*)cells + (ty*TILE_PITCH + tx);
            cell[0] = half4{(_Float16)r00, (_Float16)g00, (_Float16)b00, (_Float16)(r10 - r00)};
            cell[PLANE_CELLS] = half4{(_Float16)(g10 - g00), (_Float16)(b10 - b00), (_Float16)(r01 - r00), (_Float16)(g01 - g00)};
            cell[2*PLANE_CELLS] = half4{(_Float16)(b01 - b00), (_Float16)((r00 - r10) - (r01 - r11)), (_Float16)((g00 - g10) - (g01 - g11)), (_Float16)((b00 - b10) - (b01 - b11))};
        } else {
            float4* cell = cells + (ty*TILE_PITCH + tx)*3;
            cell[0] = make_float4(r00, g00, b00, r10 - r00);
            cell[1] = make_float4(g10 - g00, b10 - b00, r01 - r00, g01 - g00);
            cell[2] = make_float4(b01 - b00, (r00 - r10) - (r01 - r11), (g00 - g10) - (g01 - g11), (b00 - b10) - (b01 - b11));
        }
    }
}

// ---- the fused kernel ------------------------------------------------------------------------------------------------------
// S == 2: a quad of lanes per output pixel, one supersample each (render_kernels.hpp render_resolve_body's layout and epilogue).
template <int TILE_PITCH, int TILE_ROWS, int BLOCK_PX>
struct VisualizerFast {
    static constexpr int S = 2;
    static constexpr int THREADS = 4*BLOCK_PX;
    struct Shared {
        float4 cells[TILE_ROWS*TILE_PITCH*3];
        float4 row_entries[S][VIS_ENTRY_QUADS];
        uint8_t staged[BLOCK_PX*3 + 16];
    };

    __device__ __forceinline__ static void cell_sum(const char* p, int offset, float wa, float wx, float wy, float wxy, float& r, float& g, float& b) {
        const float4* q = (const float4*)(p + offset);
        const float4 q0 = q[0], q1 = q[1], q2 = q[2];
        r = fmaf(wa, q0.x, r);   g = fmaf(wa, q0.y, g);   b = fmaf(wa, q0.z, b);
        r = fmaf(wx, q0.w, r);   g = fmaf(wx, q1.x, g);   b = fmaf(wx, q1.y, b);
        r = fmaf(wy, q1.z, r);   g = fmaf(wy, q1.w, g);   b = fmaf(wy, q2.x, b);
        r = fmaf(wxy, q2.y, r);  g = fmaf(wxy, q2.z, g);  b = fmaf(wxy, q2.w, b);
    }

    // The blur without the LDS window (a block whose window does not fit the tile, or a radius beyond the line slots — backgrounds
    // much larger than the output): the 91 taps of visualizer.frag:19-31 as bilinear fetches from global memory, in the same texel
    // coordinates and difference form as the tiled taps. A compact loop: its registers must not weigh on the fast path.
    __device__ static void blur_direct(const RenderArgs& a, const Tex& bg, float x, float y, float intensity, float& r, float& g, float& b) {
        const float ax = intensity*a.bg_scale_x*(float)bg.width, ay = intensity*(float)bg.height;
        const uint8_t* data = (const uint8_t*)bg.data;
        const int comps = bg.components;
        typedef uint32_t unaligned_u32 __attribute__((aligned(1)));
#pragma unroll 1
        for (int k = 0; k <= 80; k++) {
            const float weight = (k < 10) ? 2.0f : 1.0f;                  // direction 0 is direction 8 as well
            const float tx = fmaf(a.tap_x[k], ax, x), ty = fmaf(a.tap_y[k], ay, y);
            const float cx = floorf(tx), cy = floorf(ty);
            const float fx = tx - cx, fy = ty - cy;
            const int i0 = wrap_texel((int)cx, bg.width, bg.repeat_x), i1 = wrap_texel((int)cx + 1, bg.width, bg.repeat_x);
            const int j0 = wrap_texel((int)cy, bg.height, bg.repeat_y), j1 = wrap_texel((int)cy + 1, bg.height, bg.repeat_y);
            const uint32_t row0 = (uint32_t)j0*(uint32_t)bg.width, row1 = (uint32_t)j1*(uint32_t)bg.width;
            const uint32_t w00 = *(const unaligned_u32*)(data + (size_t)(row0 + i0)*comps), w10 = *(const unaligned_u32*)(data + (size_t)(row0 + i1)*comps);
            const uint32_t w01 = *(const unaligned_u32*)(data + (size_t)(row1 + i0)*comps), w11 = *(const unaligned_u32*)(data + (size_t)(row1 + i1)*comps);
            const float wx = weight*fx, wy = weight*fy, wxy = wx*fy;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float t00 = (float)((w00 >> (8*ch)) & 255u), t10 = (float)((w10 >> (8*ch)) & 255u);
                const float t01 = (float)((w01 >> (8*ch)) & 255u), t11 = (float)((w11 >> (8*ch)) & 255u);
                float& acc = ch == 0 ? r : (ch == 1 ? g : b);
                acc = fmaf(weight, t00, acc);
                acc = fmaf(wx, t10 - t00, acc);
                acc = fmaf(wy, t01 - t00, acc);
                acc = fmaf(wxy, (t00 - t10) - (t01 - t11), acc);
            }
        }
    }

    __device__ static void run(const RenderArgs& a, const VisTables& t) {
        __shared__ __attribute__((aligned(16))) Shared sh;
        const int frame = blockIdx.z;
        const int tile_index = xcd_band_order(blockIdx.x, gridDim.x);
        const int bx = tile_index % t.blocks_x, by = tile_index / t.blocks_x;
        const int tid = threadIdx.x;
        const int p = tid >> 2, sub = tid & 3;
        const int px = bx*BLOCK_PX + p, py = by;
        const bool valid = (px < a.w) && (py < a.h);
        const int i = min(px, a.w - 1)*S + (sub & 1);

        const int4 wx = t.block_x[(long)frame*t.blocks_x + bx], wy = t.block_y[(long)frame*t.blocks_y + by];
        const Tex& bg = a.tex[TEX_BACKGROUND];
        const VisualizerConsts c = a.vis_consts ? a.vis_consts[a.frame0 + frame] : a.vis;

        // column entry: global (L1/L2 resident: 4320 sample rows read each column's entry)
        const float4* ce = t.columns + ((long)frame*a.wr + i)*VIS_ENTRY_QUADS;

        // stage the window (VisualizerShader::setup step 2) and the block's two row entries
        if (wx.z && wy.z) {
            visualizer_fast_stage<TILE_PITCH, THREADS>(bg, sh.cells, wx.x, wy.x, wx.y, wy.y, tid);
            if (tid < S*VIS_ENTRY_QUADS) {
                const int row = tid / VIS_ENTRY_QUADS, quad = tid - row*VIS_ENTRY_QUADS;
                const int jr = min(by*S + row, a.hr - 1);
                sh.row_entries[row][quad] = t.rows[((long)frame*a.hr + jr)*VIS_ENTRY_QUADS + quad];
            }
        }
        __syncthreads();
        const float4* re = sh.row_entries[sub >> 1];
        const float4 r0 = re[0], c0 = ce[0];

        uint32_t texel = 0;
        if (valid) {
            float r = 0.0f, g = 0.0f, b = 0.0f;
            const char* tile = (const char*)sh.cells;
            const float xr = c0.x, fx = c0.y, yr = r0.x, fy = r0.y;
            if (!(wx.z && wy.z)) {
                blur_direct(a, bg, xr + (float)wx.x, yr + (float)wy.x, c.intensity, r, g, b);
            } else {
            // the row of texels through the sample (directions 0, 180 degrees and the centre tap): moving axis x, fixed fraction fy
            {
                const char* line = tile + (__float_as_int(c0.w) + __float_as_int(r0.z));
                const float4 c3 = ce[3], c4 = ce[4], c5 = ce[5], c6 = ce[6];
                const float n[8] = {c3.x, c3.z, c4.x, c4.z, c5.x, c5.z, c6.x, c6.z}, s[8] = {c3.y, c3.w, c4.y, c4.w, c5.y, c5.w, c6.y, c6.w};
#pragma unroll
                for (int k = 0; k < VIS_LINE_CELLS; k++) cell_sum(line, k*48, n[k], s[k], n[k]*fy, s[k]*fy, r, g, b);
            }
            // the column of texels (90 and 270 degrees): moving axis y, fixed fraction fx
            {
                const char* line = tile + (__float_as_int(r0.w) + __float_as_int(c0.z));
                const float4 l3 = re[3], l4 = re[4], l5 = re[5], l6 = re[6];
                const float n[8] = {l3.x, l3.z, l4.x, l4.z, l5.x, l5.z, l6.x, l6.z}, s[8] = {l3.y, l3.w, l4.y, l4.w, l5.y, l5.w, l6.y, l6.w};
#pragma unroll
                for (int k = 0; k < VIS_LINE_CELLS; k++) cell_sum(line, k*TILE_PITCH*48, n[k], n[k]*fx, s[k], s[k]*fx, r, g, b);
            }
            // the four diagonal directions (VisualizerShader::blur_tile): at every walk step the four taps (+-k*s, +-k*s) share two
            // x and two y coordinates
            {
                const float ax = c.intensity*a.bg_scale_x*(float)bg.width;
                const float step = (a.tap_x[11] - a.tap_x[10])*ax, first = a.tap_x[10]*ax;
                const float ROW = (float)(TILE_PITCH*48);
                const float4* cells = sh.cells;
                float xp = xr + first, xm = xr - first, yp = yr + first, ym = yr - first;
                using V = VisualizerShader<TILE_PITCH, TILE_ROWS, 8>;
#ifndef VIS_FAST_UNROLL
#define VIS_FAST_UNROLL 1
#endif
#pragma unroll VIS_FAST_UNROLL
                for (int w = 0; w < 10; w++) {
                    const float axp = __builtin_amdgcn_fractf(xp), axm = __builtin_amdgcn_fractf(xm);
                    const float ayp = __builtin_amdgcn_fractf(yp), aym = __builtin_amdgcn_fractf(ym);
                    const float cxp = (xp - axp)*48.0f, cxm = (xm - axm)*48.0f;
                    const float ryp = (yp - ayp)*ROW, rym = (ym - aym)*ROW;
                    V::tap_at(cells, cxp + ryp, axp, ayp, r, g, b);
                    V::tap_at(cells, cxm + ryp, axm, ayp, r, g, b);
                    V::tap_at(cells, cxm + rym, axm, aym, r, g, b);
                    V::tap_at(cells, cxp + rym, axp, aym, r, g, b);
                    xp = xp + step; xm = xm - step; yp = yp + step; ym = ym - step;
                }
            }
            }
            texel = visualizer_fast_post(a, frame, c, r, g, b, ce[1], ce[2], re[1], re[2]);
        }

        // final.glsl over the pixel's 2x2 block (render_resolve_body): lane c of the quad resolves channel c
        const uint32_t l0 = quad_lane0(texel), l1 = quad_lane1(texel), l2 = quad_lane2(texel), l3 = quad_lane3(texel);
        const uint32_t block[4] = {l0, l1, l2, l3};
        const uint32_t channel = resolve_channel_any<2>(block, a.subsample, 8*(sub < 3 ? sub : 0));
        const uint32_t green = quad_lane1(channel), blue = quad_lane2(channel);
        if (valid && sub == 0) {
            uint8_t* s = &sh.staged[p*3];
            s[0] = (uint8_t)channel; s[1] = (uint8_t)green; s[2] = (uint8_t)blue;
        }
        __syncthreads();
        uint8_t* out = (uint8_t*)a.out + (long)frame*a.out_frame_stride;
        if (py < a.h) store_rgb_row(out + (long)(a.top_down ? a.h - 1 - py : py)*a.w*3, bx*BLOCK_PX, a.w, sh.staged, tid, THREADS, BLOCK_PX);
    }
};

// ---- the same, with every lane walking a strip of WALK samples of its column ---------------------------------------------------
// What a supersample shares with the ones right above it in the same column (a quarter of a texel apart at 4K 2xSSAA over a
// 1080-row background, a sixteenth at 8K 4xSSAA):
//   * the row-line: its per-cell weights (n, s) depend on the column only, and so do its cells as long as the two samples lie in
//     the same row of texel cells — then the line's sums P = sum(n*A + s*B), Q = sum(n*C + s*D) are the same and the samples
//     differ by the lerp P + fy*Q alone: 8 cell fetches and 96 multiply-adds once per distinct cell row instead of per sample;
//   * the column-line's cells: the same column of cells, shifted by at most a cell per sample — fetched once for the strip and
//     folded with the column's fraction once (U = A + fx*B, V = C + fx*D), then every sample whose 8 slots cover the cell adds
//     n*U + s*V with its own row's weights;
//   * the x half of every diagonal tap: the taps (x +- k*s, y' +- k*s) of all the strip's samples y' have the same two x
//     coordinates, so wherever two samples' taps land in the same row of cells they read the SAME two cells, and with
//     U = (A+ax*B)(x+) + (A+ax*B)(x-), V = (C+ax*D)(x+) + (C+ax*D)(x-) folded once per distinct cell row a sample's pair of taps
//     is U + ay*V: six operations for two taps instead of twenty-six, and the cells are fetched once per distinct row;
//   * the column entry and the window staging.
// For that the LANES OF A WAVE MUST SHARE THEIR ROWS (the questions "same cell row?" and "which slots cover this cell?" are then
// wave-uniform: scalar branches on values from scalar loads, no divergence), so the S x S supersamples of a pixel no longer sit in
// the four lanes of a quad: wave w holds the WALK consecutive sample rows of row group w % ROW_GROUPS for the 64 sample columns of
// column group w / ROW_GROUPS; the RGBA8 texels meet in LDS (over the cells, which are dead by then) and one thread per output pixel resolves all
// three channels — cheaper than the DPP exchange it replaces.
#ifdef SF_SECTION_TIMERS
#define SF_COUNT_FOLD() (sf_folds++)
#define SF_FOLDS_OUT(a) do { if ((threadIdx.x & 63) == 0 && (a).timers) atomicAdd(&(a).timers[(((blockIdx.x + blockIdx.z*gridDim.x)*8 + (threadIdx.x >> 6)) % SF_TIMER_ROWS)*8 + 7], (unsigned long long)sf_folds); } while (0)
#else
#define SF_COUNT_FOLD() do {} while (0)
#define SF_FOLDS_OUT(a) do {} while (0)
#endif
// (Round 6: the measured dead ends that used to sit here as VIS_STRIP_* build knobs — adjacent-row folds, the fractions through LDS or
// vector memory, the two sides' U summed once, blue packed into a split accumulator, the rows' scalar loads issued together, a store
// per row instead of the sweep — are gone from the source; what each cost is profiles/r05_variants.txt and HISTORY.)
template <int TILE_PITCH, int TILE_ROWS, int S, int WALK, int COLUMN_GROUPS = 8/S, bool HALF_CELLS = (S == 1)>
struct VisualizerStrip {
    // S x S supersamples per pixel (2 or 4), or S == 1: no resolve, the RGBA8 samples go to iScreen (the two-pass configuration).
    // A block is 512 threads = 8 waves = COLUMN_GROUPS groups of 64 sample columns x ROW_GROUPS groups of WALK consecutive rows.
    static constexpr int THREADS = 512;
    static constexpr int ROW_GROUPS = 8/COLUMN_GROUPS;
    static constexpr int COLS = 64*COLUMN_GROUPS;                      // sample columns of a block: 256 (S = 2), 128 (S = 4), 64 (S = 1)
    static constexpr int BLOCK_PX = COLS/S;                            // output pixels of a row per block: 128, 32
    static constexpr int RROWS = ROW_GROUPS*WALK;                      // sample rows of a block
    static constexpr int PIXEL_ROWS = RROWS/S;                         // output rows of a block
    // Cell format. At 4K 2xSSAA the lanes of a wave are a quarter of a texel apart and mostly read the SAME cell (a broadcast): the
    // kernel is bound by VALU issue and float32 cells with plain v_fmac are the cheapest. Without SSAA at 1080p they
    // are 0.87 texel apart: 56 DIFFERENT cells per wave, every ds_read_b128 moves 1 KB = 8 LDS cycles, and the kernel is bound by LDS
    // bandwidth (SQ_LDS_IDX_ACTIVE 0.90 of the CU cycles, VALU 0.55; profiles/r03_c2_counters.txt). There the cells are FLOAT16 —
    // every component is a texel byte or a difference of texel bytes, an integer of magnitude <= 510: exact — in three planes of 8
    // bytes per cell: half the LDS bytes, the multiply-adds that touch cell data become v_fma_mix_f32 (f16 operands converted on the
    // fly, half issue rate — the slack is there), and the results are the float32 kernel's bit for bit.
    static constexpr bool HALF = HALF_CELLS;
    using Quad = typename std::conditional<HALF, half4, float4>::type;
    static constexpr int CELL = HALF ? 8 : 48;                         // bytes between neighbouring cells
    static constexpr int PLANE = HALF ? TILE_ROWS*TILE_PITCH*8 : 16;   // bytes between the three quads of one cell
    static_assert(2*PLANE < 65536, "the planes are reached through ds_read's 16-bit immediate offset");
    static constexpr int ROWBYTES = TILE_PITCH*CELL;
    __device__ __forceinline__ static void load_cell(const char* p, Quad& q0, Quad& q1, Quad& q2) {
        q0 = *(const Quad*)p; q1 = *(const Quad*)(p + PLANE); q2 = *(const Quad*)(p + 2*PLANE);
    }
    template <class T> __device__ __forceinline__ static float F(T x) { return (float)x; }     // a cell component as a multiply-add operand
    static_assert(COLUMN_GROUPS*ROW_GROUPS == 8 && RROWS % S == 0 && COLS % S == 0, "block geometry");
    using Fast = VisualizerFast<TILE_PITCH, TILE_ROWS, 128>;
    // the cell tile; once every wave is done with it, the texel exchange and the staged RGB8 rows live in the same memory
    static constexpr int CELLS_BYTES = TILE_ROWS*TILE_PITCH*(HALF_CELLS ? 24 : 48);
    static constexpr int EXCHANGE_BYTES = (S == 1) ? 0 : (int)sizeof(uint32_t)*ROW_GROUPS*WALK*64*COLUMN_GROUPS + (ROW_GROUPS*WALK/S)*(64*COLUMN_GROUPS/S)*3;
    // Round 5 (profiles/r05_ubench_valu_sgpr.txt): a VALU instruction with a SCALAR source — v_mov_b32 v, s included — issues in 4.2
    // cycles, not 2.5. The column-line's wave-uniform weights therefore arrive in vector registers through LDS reads at an address held
    // in ONE vector register plus an immediate offset — no VALU instruction at all: per WAVE, aligned to the strip's first cell and zero
    // padded: ds_read_b64 at v_k + r*COLW_SLOTS*8
    static constexpr int COLW_SLOTS = VIS_LINE_CELLS + 6;              // cells a strip's column can span: the 8 slots of a line + the rows of cells its samples cross
    struct Shared {
        float4 cells[(CELLS_BYTES > EXCHANGE_BYTES ? CELLS_BYTES : EXCHANGE_BYTES)/16 + 1];     // (float16 cells without SSAA) later: uint32 texels[RROWS][COLS], then the RGB8 rows at STAGED
        float4 row_entries[RROWS][VIS_ENTRY_QUADS];
        float4 zeros;                                                  // weights of a slot that does not exist
        float2 colw[8][WALK][COLW_SLOTS];                              // per wave: weights[r][k - first]
    };
    static constexpr int STAGED = (int)sizeof(uint32_t)*RROWS*COLS;    // byte offset of the staged RGB8 rows inside the (dead) cell tile, after the texels
    static_assert(S == 1 || STAGED + PIXEL_ROWS*BLOCK_PX*3 == EXCHANGE_BYTES, "the texel exchange and the staged rows live in the cell tile");

    // The pixel tier of this wave (visualizer_pixel_gains): OFFSET = the wave's first sample row modulo S. Fills texel[] for every row.
    // The S lanes that hold the S sample columns of one pixel column SHARE the work: in round j lane l evaluates the gains of pixel row
    // S*j + l % S of the wave's SLOTS pixel rows (at 2x: five pixel rows in three rounds, at 4x: up to four in one), and a sample takes its
    // pixel's pair (A, w) from the lane that evaluated it through a quad_perm broadcast — DPP operands of the instructions that use them.
    template <int SOURCE> __device__ __forceinline__ static float from_pixel_lane(float v) {
        constexpr int CTRL = S == 2 ? (SOURCE | (SOURCE << 2) | ((2 + SOURCE) << 4) | ((2 + SOURCE) << 6)) : SOURCE*0x55;    // quad_perm: [s, s, 2+s, 2+s] / [s, s, s, s]
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
    }
    template <int OFFSET>
    __device__ __forceinline__ static void pixel_tier(const RenderArgs& a, const VisTables& t, int frame, const VisualizerConsts& c, const float (&acc)[WALK][3], uint32_t (&texel)[WALK],
                                                      const float4 c2, const Shared& sh, int bx, int by, int row0, int rows, int column) {
        static_assert(S == 2 || S == 4, "the lanes of a pixel column are a pair or a quad");
        constexpr int SLOTS = (OFFSET + WALK - 1)/S + 1;              // pixel rows this wave's sample rows fall into
        constexpr int ROUNDS = (SLOTS + S - 1)/S;
        const int pixel_column = min((bx*COLS + column)/S, a.w - 1);
        float4 pc1 = t.pixel_columns[(long)frame*a.w + pixel_column];
        const int pixel_row0 = by*PIXEL_ROWS + row0/S + (column % S);  // this lane's pixel row of round 0
        // The waveform strips (:72-73) scale a sample by 0.8 each: a factor of its own per sample, evaluated only by the waves that can see a
        // strip — the rows of a wave are consecutive, 1 - gluv.y falls and 1 + gluv.y rises with the row, so the wave's last and first rows
        // decide for all of them
        const float4 r2_first = sh.row_entries[row0][2], r2_last = sh.row_entries[row0 + (rows > 0 ? rows - 1 : 0)][2];
        const bool strips = __builtin_amdgcn_ballot_w64((r2_last.y < c2.y) || (r2_first.z < c2.z)) != 0;
        PixelGains gains[ROUNDS];
#pragma unroll
        for (int j = 0; j < ROUNDS; j++) {
            const int pixel_row = pixel_row0 + S*j;
            const float4 pr1 = t.pixel_rows[(long)frame*a.h + (pixel_row < a.h ? pixel_row : a.h - 1)];
            gains[j] = visualizer_pixel_gains(a, t, frame, c, pc1, pr1);
            // the rounds ONE AFTER THE OTHER: left alone the scheduler interleaves the independent evaluations and their live values no longer
            // fit the 80 registers — an empty asm makes the next round's first input depend on this round's result (plain asm, not volatile:
            // see the column lines)
            asm("" : "+v"(pc1.x) : "v"(gains[j].w));
        }
        auto apply = [&](int r, float A, float w) {
            const float4 r2 = sh.row_entries[row0 + r][2];
            float scale = (c2.x*r2.x)*A;                              // the vignette's two factors and the pixel's gain
            if (strips) {
                if (r2.y < c2.y) scale = scale*0.8f;                  // :72
                if (r2.z < c2.z) scale = scale*0.8f;                  // :73
            }
            const float v0 = fmaf(1.0f, w, acc[r][0])*scale, v1 = fmaf(11.0f, w, acc[r][1])*scale, v2 = fmaf(26.0f, w, acc[r][2])*scale;
            uint32_t packed = __builtin_amdgcn_cvt_pk_u8_f32(v0, 0u, 0u);
            packed = __builtin_amdgcn_cvt_pk_u8_f32(v1, 1u, packed);
            texel[r] = __builtin_amdgcn_cvt_pk_u8_f32(v2, 2u, packed);
        };
#pragma unroll
        for (int r = 0; r < WALK; r++) {
            constexpr int dummy = 0; (void)dummy;
            const int p = (OFFSET + r)/S;                             // (constants after unrolling)
            const PixelGains& g = gains[p/S];
            switch (p % S) {
                case 0: apply(r, from_pixel_lane<0>(g.A), from_pixel_lane<0>(g.w)); break;
                case 1: apply(r, from_pixel_lane<1>(g.A), from_pixel_lane<1>(g.w)); break;
                case 2: apply(r, from_pixel_lane<2 % S>(g.A), from_pixel_lane<2 % S>(g.w)); break;
                default: apply(r, from_pixel_lane<3 % S>(g.A), from_pixel_lane<3 % S>(g.w)); break;
            }
        }
    }

    __device__ static void run(const RenderArgs& a, const VisTables& t) {
        __shared__ __attribute__((aligned(16))) Shared sh;
        SF_TICK_INIT();                                               // profiling builds (-DSF_SECTION_TIMERS): wave cycles per phase, see capi.hip
#ifdef SF_SECTION_TIMERS
        int sf_folds = 0;                                             // diagonal cell folds of this wave (wave-uniform branches)
#endif
        const int frame = blockIdx.z;
        const int tile_index = xcd_band_order(blockIdx.x, gridDim.x);
        // (the division runs on the vector unit: say that its results are the same in every lane)
        const int bx = __builtin_amdgcn_readfirstlane(tile_index % t.blocks_x), by = __builtin_amdgcn_readfirstlane(tile_index / t.blocks_x);
        const int tid = threadIdx.x;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const int group = wave % ROW_GROUPS;                          // row group of this wave: sample rows group*WALK + r of the block
        const int column = (wave / ROW_GROUPS)*64 + lane;             // sample column inside the block
        const int i = (bx*COLS + column < a.wr) ? bx*COLS + column : a.wr - 1;
        const int row0 = group*WALK;                                  // first row of the wave inside the block
        const int jr0 = by*RROWS + row0;                              // and inside the frame
        // sample rows of this wave that exist (the frame's height need not be a multiple of the block's)
        const int rows = (a.hr - jr0 < WALK) ? (a.hr - jr0 < 0 ? 0 : a.hr - jr0) : WALK;

        const int4 wx = t.block_x[(long)frame*t.blocks_x + bx], wy = t.block_y[(long)frame*t.blocks_y + by];
        const bool fits = wx.z && wy.z;
        const Tex& bg = a.tex[TEX_BACKGROUND];
        const VisualizerConsts c = a.vis_consts ? a.vis_consts[a.frame0 + frame] : a.vis;
        const float4* ce = t.columns + ((long)frame*a.wr + i)*VIS_ENTRY_QUADS;

        if (fits) visualizer_fast_stage<TILE_PITCH, THREADS, HALF ? TILE_ROWS*TILE_PITCH : 0>(bg, sh.cells, wx.x, wy.x, wx.y, wy.y, tid);
        if (tid < RROWS*VIS_ENTRY_QUADS) {
            const int row = tid / VIS_ENTRY_QUADS, quad = tid - row*VIS_ENTRY_QUADS;
            const int jr = by*RROWS + row;
            sh.row_entries[row][quad] = t.rows[((long)frame*a.hr + (jr < a.hr ? jr : a.hr - 1))*VIS_ENTRY_QUADS + quad];
        }
        if (tid == 0) sh.zeros = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        __syncthreads();
        SF_TICK(a, 0);                                                // prologue: tables, staging the cell tile, the first barrier

        const float4 c0 = ce[0];
        const float xr = c0.x, fx = c0.y;
        const char* tile = (const char*)sh.cells;
        float acc[WALK][3];
#pragma unroll
        for (int r = 0; r < WALK; r++) { acc[r][0] = 0.0f; acc[r][1] = 0.0f; acc[r][2] = 0.0f; }

        if (!fits) {
#pragma unroll
            for (int r = 0; r < WALK; r++)
                if (r < rows) Fast::blur_direct(a, bg, xr + (float)wx.x, sh.row_entries[row0 + r][0].x + (float)wy.x, c.intensity, acc[r][0], acc[r][1], acc[r][2]);
        } else {
            // ---- the row-lines: P and Q once per distinct row of cells ----
            {
                const float4 c3 = ce[3], c4 = ce[4], c5 = ce[5], c6 = ce[6];
                const float n[8] = {c3.x, c3.z, c4.x, c4.z, c5.x, c5.z, c6.x, c6.z}, s[8] = {c3.y, c3.w, c4.y, c4.w, c5.y, c5.w, c6.y, c6.w};
                float P[3] = {0.0f, 0.0f, 0.0f}, Q[3] = {0.0f, 0.0f, 0.0f};
                int previous = -2*ROWBYTES;                           // (no row of cells: neither equal nor adjacent to a real one)
#pragma unroll
                for (int r = 0; r < WALK; r++) if (r < rows) {
                    const float4 r0 = sh.row_entries[row0 + r][0];
                    const int cell_row = __builtin_amdgcn_readfirstlane(__float_as_int(r0.z));       // the wave's lanes share their rows
                    if (cell_row != previous) {
                        previous = cell_row;
                        const char* line = tile + (__float_as_int(c0.w) + cell_row);
                        P[0] = P[1] = P[2] = Q[0] = Q[1] = Q[2] = 0.0f;
#pragma unroll
                        for (int k = 0; k < VIS_LINE_CELLS; k++) {
                            Quad q0, q1, q2;
                            load_cell(line + k*CELL, q0, q1, q2);
                            P[0] = fmaf(n[k], F(q0.x), P[0]); P[1] = fmaf(n[k], F(q0.y), P[1]); P[2] = fmaf(n[k], F(q0.z), P[2]);
                            P[0] = fmaf(s[k], F(q0.w), P[0]); P[1] = fmaf(s[k], F(q1.x), P[1]); P[2] = fmaf(s[k], F(q1.y), P[2]);
                            Q[0] = fmaf(n[k], F(q1.z), Q[0]); Q[1] = fmaf(n[k], F(q1.w), Q[1]); Q[2] = fmaf(n[k], F(q2.x), Q[2]);
                            Q[0] = fmaf(s[k], F(q2.y), Q[0]); Q[1] = fmaf(s[k], F(q2.z), Q[1]); Q[2] = fmaf(s[k], F(q2.w), Q[2]);
                        }
                    }
                    acc[r][0] = fmaf(r0.y, Q[0], P[0]); acc[r][1] = fmaf(r0.y, Q[1], P[1]); acc[r][2] = fmaf(r0.y, Q[2], P[2]);
                }
            }
            SF_TICK(a, 1);                                            // row-lines
            // ---- the column-lines: the strip's column of cells fetched and folded with fx once, every sample takes the cells its slots cover ----
            {
                int start[WALK];
                int last = 0;
#pragma unroll
                for (int r = 0; r < WALK; r++) {
                    start[r] = __builtin_amdgcn_readfirstlane(__float_as_int(sh.row_entries[row0 + r][0].w))/ROWBYTES;
                    if (r < rows) last = start[r] + VIS_LINE_CELLS;     // start[] does not decrease with the row
                }
                const int first = start[0];
                const char* column_cells = tile + __float_as_int(c0.z);
                // (a tile of at most COLW_SLOTS rows of cells cannot hold a longer run: the other loop is then not even compiled, and with it
                // go the 2 x 27 v_mov that carried the accumulators into and out of the registers the two loops disagreed about)
                if (TILE_ROWS <= COLW_SLOTS || last - first <= COLW_SLOTS) {
                    // this wave's weights, aligned to the strip's first cell: colw[r][j] = (n, s) of row r for cell first + j, zeros where
                    // the row's eight slots do not reach. Built by the wave itself (LDS operations of one wave execute in order: no barrier)
                    float2* mine = &sh.colw[wave][0][0];
#pragma unroll
                    for (int e = lane; e < WALK*COLW_SLOTS; e += 64) {
                        const int r = e / COLW_SLOTS, j = e - r*COLW_SLOTS;
                        const int begin = __float_as_int(sh.row_entries[row0 + r][0].w)/ROWBYTES;
                        const int slot = j + first - begin;
                        const bool covered = r < rows && slot >= 0 && slot < VIS_LINE_CELLS;
                        mine[e] = covered ? *(const float2*)((const char*)sh.row_entries[row0 + r] + 48 + slot*8) : make_float2(0.0f, 0.0f);
                    }
                    int vzero;
                    asm("v_mov_b32 %0, 0" : "=v"(vzero));                   // the table's address lives in a VECTOR register: ds_read_b64 v, v_k offset:r*COLW_SLOTS*8
                    // (NOT `asm volatile`: a side-effecting asm counts as a possible store, after which the compiler no longer proves the
                    // per-frame tables unclobbered and fetches them with vector instead of scalar loads — 78 -> 100 registers)
                    const char* weights_k = (const char*)mine + vzero;
                    const char* cell_k = column_cells + first*ROWBYTES;
                    float U0 = 0.0f, U1 = 0.0f, U2 = 0.0f, V0 = 0.0f, V1 = 0.0f, V2 = 0.0f;
                    for (int k = first; k < last; k++) {
                        Quad q0, q1, q2;
                        load_cell(cell_k, q0, q1, q2);
                        U0 = fmaf(fx, F(q0.w), F(q0.x)); U1 = fmaf(fx, F(q1.x), F(q0.y)); U2 = fmaf(fx, F(q1.y), F(q0.z));
                        V0 = fmaf(fx, F(q2.y), F(q1.z)); V1 = fmaf(fx, F(q2.z), F(q1.w)); V2 = fmaf(fx, F(q2.w), F(q2.x));
#pragma unroll
                        for (int r = 0; r < WALK; r++) {
                            const float2 w = *(const float2*)(weights_k + r*COLW_SLOTS*8);
                            acc[r][0] = fmaf(w.x, U0, acc[r][0]); acc[r][1] = fmaf(w.x, U1, acc[r][1]); acc[r][2] = fmaf(w.x, U2, acc[r][2]);
                            acc[r][0] = fmaf(w.y, V0, acc[r][0]); acc[r][1] = fmaf(w.y, V1, acc[r][1]); acc[r][2] = fmaf(w.y, V2, acc[r][2]);
                        }
                        weights_k += 8; cell_k += ROWBYTES;
                    }
                } else
                for (int k = first; k < last; k++) {
                    // a run of cells longer than the aligned table (tiles taller than COLW_SLOTS rows only): the weights of all rows of a cell
                    // at once, zeros for rows whose slots do not cover it
                    Quad q0, q1, q2;
                    load_cell(column_cells + k*ROWBYTES, q0, q1, q2);
                    const float U0 = fmaf(fx, F(q0.w), F(q0.x)), U1 = fmaf(fx, F(q1.x), F(q0.y)), U2 = fmaf(fx, F(q1.y), F(q0.z));
                    const float V0 = fmaf(fx, F(q2.y), F(q1.z)), V1 = fmaf(fx, F(q2.z), F(q1.w)), V2 = fmaf(fx, F(q2.w), F(q2.x));
                    float2 weights[WALK];
#pragma unroll
                    for (int r = 0; r < WALK; r++) {
                        const int slot = k - start[r];
                        const bool covered = r < rows && slot >= 0 && slot < VIS_LINE_CELLS;
                        const char* source = covered ? (const char*)sh.row_entries[row0 + r] + 48 + slot*8 : (const char*)&sh.zeros;
                        weights[r] = *(const float2*)source;
                    }
#pragma unroll
                    for (int r = 0; r < WALK; r++) {
                        const float2 w = weights[r];
                        acc[r][0] = fmaf(w.x, U0, acc[r][0]); acc[r][1] = fmaf(w.x, U1, acc[r][1]); acc[r][2] = fmaf(w.x, U2, acc[r][2]);
                        acc[r][0] = fmaf(w.y, V0, acc[r][0]); acc[r][1] = fmaf(w.y, V1, acc[r][1]); acc[r][2] = fmaf(w.y, V2, acc[r][2]);
                    }
                }
            }
            SF_TICK(a, 2);                                            // column-lines
            // ---- the four diagonal directions: per walk step the x halves once for the strip; the y halves (fraction and row of
            //      cells of y +- k*s, k_visualizer_axis<1>'s ysteps) are wave-uniform: scalar loads, scalar branches ----
            {
                const float ax = c.intensity*a.bg_scale_x*(float)bg.width;
                const float step = (a.tap_x[11] - a.tap_x[10])*ax, first = a.tap_x[10]*ax;
                const float4* ysteps[WALK];
#pragma unroll
                for (int r = 0; r < WALK; r++) {
                    const int jr = jr0 + r;
                    ysteps[r] = t.ysteps + ((long)frame*a.hr + (jr < a.hr ? jr : a.hr - 1))*10;
                }
                float xp = xr + first, xm = xr - first;
#pragma unroll 1
                for (int w = 0; w < 10; w++) {
                    const float axp = __builtin_amdgcn_fractf(xp), axm = __builtin_amdgcn_fractf(xm);
                    const int cxp = (int)((xp - axp)*(float)CELL), cxm = (int)((xm - axm)*(float)CELL);
                    float4 y[WALK];
#pragma unroll
                    for (int r = 0; r < WALK; r++) y[r] = ysteps[r][w];   // { frac(y+), frac(y-), row bytes(y+), row bytes(y-) }
                    float U[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}}, V[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
                    int previous[2] = {-2*ROWBYTES, -2*ROWBYTES};     // (no row of cells: neither equal nor adjacent to a real one)
                    // both y sides of a walk step advance row by row together (their first cells are fetched at once)
                    auto advance = [&](int r, int side) {
                        const int cell_row = __float_as_int(side ? y[r].w : y[r].z);
                        // (a walk step's first row folds without asking — it always does: `previous` starts at no row — so that the compiler drops
                        // the thirteen v_mov 0 that initialised U and V for the path nobody takes)
                        if (r == 0 || cell_row != previous[side]) {
                            previous[side] = cell_row;
                            SF_COUNT_FOLD();
                            Quad p0, p1, p2, m0, m1, m2;
                            load_cell(tile + (cxp + cell_row), p0, p1, p2);
                            load_cell(tile + (cxm + cell_row), m0, m1, m2);
                            float* u = U[side]; float* v = V[side];
                            if constexpr (HALF) {
                                // the sums of two components are exact in float16: packed adds, two per instruction
                                const Quad s0 = p0 + m0, s1 = p1 + m1, s2 = p2 + m2;
                                u[0] = F(s0.x); u[1] = F(s0.y); u[2] = F(s0.z);
                                v[0] = F(s1.z); v[1] = F(s1.w); v[2] = F(s2.x);
                            } else {
                                // (six scalar adds: a vector add would become v_pk_add_f32 — twice the issue cost for twelve sums, six unused)
                                u[0] = p0.x + m0.x; u[1] = p0.y + m0.y; u[2] = p0.z + m0.z;
                                v[0] = p1.z + m1.z; v[1] = p1.w + m1.w; v[2] = p2.x + m2.x;
                            }
                            u[0] = fmaf(axp, F(p0.w), u[0]);  u[1] = fmaf(axp, F(p1.x), u[1]);  u[2] = fmaf(axp, F(p1.y), u[2]);
                            u[0] = fmaf(axm, F(m0.w), u[0]);  u[1] = fmaf(axm, F(m1.x), u[1]);  u[2] = fmaf(axm, F(m1.y), u[2]);
                            v[0] = fmaf(axp, F(p2.y), v[0]);  v[1] = fmaf(axp, F(p2.z), v[1]);  v[2] = fmaf(axp, F(p2.w), v[2]);
                            v[0] = fmaf(axm, F(m2.y), v[0]);  v[1] = fmaf(axm, F(m2.z), v[1]);  v[2] = fmaf(axm, F(m2.w), v[2]);
                        }
                        // red and green of a row advance together as v_pk_add_f32 / v_pk_fma_f32 with the row's SCALAR fractions broadcast by
                        // op_sel (tools/ubench_pk_f32.hip: a scalar pair costs a packed form nothing, a v_mov v, s 4.2 cycles); blue keeps
                        // scalar-source fmas
                        typedef float pk2 __attribute__((ext_vector_type(2)));
                        pk2 a01 = {acc[r][0], acc[r][1]};
                        const pk2 u01 = {U[side][0], U[side][1]}, v01 = {V[side][0], V[side][1]};
                        const pk2 fractions = {y[r].x, y[r].y};               // a scalar pair: { frac(y+), frac(y-) }
                        asm("v_pk_add_f32 %0, %0, %1" : "+v"(a01) : "v"(u01));
                        if (side == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a01) : "s"(fractions), "v"(v01));
                        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a01) : "s"(fractions), "v"(v01));
                        acc[r][0] = a01.x; acc[r][1] = a01.y;
                        acc[r][2] = acc[r][2] + U[side][2];
                        acc[r][2] = fmaf(side ? y[r].y : y[r].x, V[side][2], acc[r][2]);
                    };
#pragma unroll
                    for (int r = 0; r < WALK; r++) if (r < rows) { advance(r, 0); advance(r, 1); }
                    xp = xp + step; xm = xm - step;
                }
            }
        }

        SF_TICK(a, 3);                                                // diagonals
        // ---- visualizer.frag:36-73 per sample, then the texels meet in LDS ----
        uint32_t texel[WALK];
        {
            const float4 c2 = ce[2];
            bool per_sample = true;
            if constexpr (S >= 2) {
                // the pixel tier for the waves whose tile k_visualizer_classify cleared — unless a lane's column lies outside the wanted aspect (:11-14)
                if (t.pixel_columns && t.wave_classes[((long)frame*gridDim.x + tile_index)*8 + wave] != 0 && __builtin_amdgcn_ballot_w64(__float_as_int(c2.w) != 0) == 0) {
                    // which output pixel a sample row belongs to depends on the wave's first row modulo S — a constant per instance of the tier,
                    // so that every index below is one
                    if constexpr ((WALK % S) == 0) pixel_tier<0>(a, t, frame, c, acc, texel, c2, sh, bx, by, row0, rows, column);
                    else {
                        const int offset = row0 % S;
                        if (offset == 0) pixel_tier<0>(a, t, frame, c, acc, texel, c2, sh, bx, by, row0, rows, column);
                        else if (offset == 1) pixel_tier<1 % S>(a, t, frame, c, acc, texel, c2, sh, bx, by, row0, rows, column);
                        else if (offset == 2) pixel_tier<2 % S>(a, t, frame, c, acc, texel, c2, sh, bx, by, row0, rows, column);
                        else pixel_tier<3 % S>(a, t, frame, c, acc, texel, c2, sh, bx, by, row0, rows, column);
                    }
                    per_sample = false;
                } else if (t.pixel_columns && a.tile_misses && (threadIdx.x & 63) == 0) atomicAdd(a.tile_misses, 1u);      // (diagnostics: sfx_ctx_tile_misses counts the WAVES the classification sent to the per-sample path)
            }
            if (per_sample) {
                const float4 c1 = ce[1];
#pragma unroll
                for (int r = 0; r < WALK; r++) {
                    texel[r] = 0;
                    if (r < rows) texel[r] = visualizer_fast_post<S == 1>(a, frame, c, acc[r][0], acc[r][1], acc[r][2], c1, c2, sh.row_entries[row0 + r][1], sh.row_entries[row0 + r][2]);
                }
            }
        }
        if constexpr (S == 1) {
            // the two-pass configuration: iScreen takes the samples as they are (alpha 1, visualizer.frag's fragColor), 64 lanes = 256 B of a row
            uint32_t* screen = (uint32_t*)((char*)a.out + (long)frame*a.out_frame_stride);
            const bool inside = bx*COLS + column < a.wr;
#pragma unroll
            for (int r = 0; r < WALK; r++)
                if (r < rows && inside) screen[(long)(jr0 + r)*a.wr + i] = texel[r];
            return;
        }
        SF_TICK(a, 4);                                                // post-processing
        __syncthreads();                                              // every wave is done with the cells
        uint32_t* texels = (uint32_t*)sh.cells;                       // [RROWS][COLS]
        uint8_t* staged = (uint8_t*)sh.cells + STAGED;                 // [PIXEL_ROWS][BLOCK_PX*3]
#pragma unroll
        for (int r = 0; r < WALK; r++) texels[(row0 + r)*COLS + column] = texel[r];
        __syncthreads();
        // final.glsl (render_kernels.hpp resolve_channel_any): one thread per output pixel, texel order y*S + x
        for (int e = tid; e < PIXEL_ROWS*BLOCK_PX; e += THREADS) {
            const int r = e / BLOCK_PX, q = e - r*BLOCK_PX;
            uint32_t block[S*S];
#pragma unroll
            for (int y = 0; y < S; y++) {
                const uint32_t* row = &texels[(S*r + y)*COLS + S*q];
                if constexpr (S == 2) { const uint2 v = *(const uint2*)row; block[y*2] = v.x; block[y*2 + 1] = v.y; }
                else { const uint4 v = *(const uint4*)row; block[y*4] = v.x; block[y*4 + 1] = v.y; block[y*4 + 2] = v.z; block[y*4 + 3] = v.w; }
            }
            uint8_t* s = staged + (r*BLOCK_PX + q)*3;
            s[0] = (uint8_t)resolve_channel_any<S>(block, a.subsample, 0);
            s[1] = (uint8_t)resolve_channel_any<S>(block, a.subsample, 8);
            s[2] = (uint8_t)resolve_channel_any<S>(block, a.subsample, 16);
        }
        __syncthreads();
        SF_TICK(a, 5);                                                // barrier + texel exchange + resolve + barrier
        SF_FOLDS_OUT(a);
        uint8_t* out = (uint8_t*)a.out + (long)frame*a.out_frame_stride;
        // a full-width block of a frame whose rows are whole 16-byte groups: all rows leave in ONE sweep of 16-byte stores by as many
        // threads as there are groups (216 at 2x) — the block ends a store latency after its resolve instead of nine
        constexpr int GROUPS = BLOCK_PX*3/16;
        if (BLOCK_PX*3 % 16 == 0 && STAGED % 16 == 0 && bx*BLOCK_PX + BLOCK_PX <= a.w && (a.w*3) % 16 == 0 && ((uintptr_t)out & 15) == 0) {
            const int rows_here = min(PIXEL_ROWS, a.h - by*PIXEL_ROWS);
#pragma unroll
            for (int e = tid; e < PIXEL_ROWS*GROUPS; e += THREADS) {
                const int r = e/GROUPS, c = e - r*GROUPS;
                if (r < rows_here) {
                    const int py = by*PIXEL_ROWS + r;
                    uint8_t* row = out + (long)(a.top_down ? a.h - 1 - py : py)*a.w*3 + (long)bx*BLOCK_PX*3;
                    stream_store16((uint4*)row + c, (const uint4*)staged + e);
                }
            }
            SF_TICK(a, 6);                                            // the sweep of stores
            return;
        }
#pragma unroll
        for (int r = 0; r < PIXEL_ROWS; r++) {
            const int py = by*PIXEL_ROWS + r;
            if (py < a.h) store_rgb_row(out + (long)(a.top_down ? a.h - 1 - py : py)*a.w*3, bx*BLOCK_PX, a.w, staged + r*BLOCK_PX*3, tid, THREADS, BLOCK_PX);
        }
    }
};

template <int TILE_PITCH, int TILE_ROWS, int S, int WALK, int MIN_WAVES, int COLUMN_GROUPS = 8/S, bool HALF_CELLS = (S == 1)>
__global__ __launch_bounds__(512, MIN_WAVES) void k_visualizer_strip(const RenderArgs a, const VisTables t) {
    VisualizerStrip<TILE_PITCH, TILE_ROWS, S, WALK, COLUMN_GROUPS, HALF_CELLS>::run(a, t);
}

template <int TILE_PITCH, int TILE_ROWS, int BLOCK_PX, int MIN_WAVES>
__global__ __launch_bounds__(4*BLOCK_PX, MIN_WAVES) void k_visualizer_fast(const RenderArgs a, const VisTables t) {
    VisualizerFast<TILE_PITCH, TILE_ROWS, BLOCK_PX>::run(a, t);
}

}  // namespace sf
