// launch_visualizer_strip.hip — the table-driven visualizer kernels (visualizer_fast.hpp): k_visualizer_axes + one instance of
// k_visualizer_strip / k_visualizer_fast per block geometry. One of the launch units of libshaderflow_hip.so (launch.hpp).
#define SF_UNIT_VISUALIZER_TABLES
#include "launch.hpp"
#include "launch_geometry.hpp"
#include "visualizer_fast.hpp"

#include <cstdlib>

using namespace sf;

namespace sfl {

// the tape's per-frame constants and bar heights (visualizer_kernels.hpp): launched from capi.hip's sfx_render_tape as well
void visualizer_consts_frames(const FrameDyn* dyn, int frame0, int nframes, VisualizerConsts* out, hipStream_t s) {
    hipLaunchKernelGGL(k_visualizer_consts, dim3((nframes + 63)/64), dim3(64), 0, s, dyn, frame0, nframes, out);
}
void visualizer_bars(const float* columns, long count, float* bars, hipStream_t s) {
    hipLaunchKernelGGL(k_visualizer_bars, dim3((unsigned)((count + 255)/256)), dim3(256), 0, s, columns, count, bars);
}

// One geometry of the fast path: the tables for it, then the kernel. STRIP_S == 0: k_visualizer_fast (a quad of lanes per pixel,
// 2x SSAA); otherwise k_visualizer_strip<PITCH, ROWS, STRIP_S, WALK, WAVES, CG> (lanes walk strips of their column; 2x or 4x SSAA
// fused with the resolve, or STRIP_S == 1: the samples to an RGBA8 iScreen).
template <int PITCH, int ROWS, int STRIP_S, int WALK, int WAVES, int CG = (STRIP_S ? 8/STRIP_S : 4), bool HALF = (STRIP_S == 1)>
static int launch_visualizer_tables_and_kernel(Context* ctx, const RenderArgs& a, int frames, hipStream_t s) {
    constexpr int COLS = STRIP_S ? 64*CG : 256;                       // sample columns per block
    constexpr int BLOCK_ROWS = STRIP_S ? (8/CG)*WALK : 2;             // sample rows per block
    VisTables t;
    t.blocks_x = (a.wr + COLS - 1)/COLS; t.blocks_y = (a.hr + BLOCK_ROWS - 1)/BLOCK_ROWS;
    t.block_columns = COLS; t.block_rows = BLOCK_ROWS; t.tile_pitch = PITCH; t.tile_rows = ROWS;
    t.cell_bytes = HALF ? 8 : 48;                                     // float16 cells in three planes where the lanes of a wave read different cells (visualizer_fast.hpp)
    const size_t entries = (size_t)frames*((size_t)a.wr + a.hr)*VIS_ENTRY_QUADS*sizeof(float4);
    const size_t blocks = (size_t)frames*((size_t)t.blocks_x + t.blocks_y)*sizeof(int4);
    const size_t ysteps = STRIP_S ? (size_t)frames*a.hr*10*sizeof(float4) : 0;
    // the pixel tier (visualizer_fast.hpp visualizer_pixel_gains): fused 2x / 4x launches whose bars come as a table; SHADERFLOW_VIS_PIXEL_TIER=0
    // turns it off (A/B measurements, and the tests that compare the two tiers)
    const Tex& sp = a.tex[TEX_SPECTROGRAM];
    const char* tier_env = getenv("SHADERFLOW_VIS_PIXEL_TIER");
    float reach_uv = 0.0f, reach_agluv = 0.0f;
    bool tier = STRIP_S >= 2 && a.tape_bars && sp.height > 0 && !(tier_env && atoi(tier_env) == 0);
    if (tier) {
        // how far a supersample lies from its pixel's centre: (S - 1)/(2S) of the pixel per axis, bounded by 0.3 of its diagonal for S <= 4
        // (0.25 at 2x, 0.375 at 4x … of the HALF diagonal 0.5): in iCamera.gluv units through the axis camera's slopes, and in agluv units
        float slope_x = 1.0f, slope_y = 1.0f;
        if (!a.identity_camera) {
            bool behind = false;
            slope_x = fabsf(camera_along_axis<0>(a.u, 1.0f, a.aspect, behind) - camera_along_axis<0>(a.u, -1.0f, a.aspect, behind))/2.0f;
            slope_y = fabsf(camera_along_axis<1>(a.u, 1.0f, a.aspect, behind) - camera_along_axis<1>(a.u, -1.0f, a.aspect, behind))/2.0f;
        }
        const float px = 2.0f*a.aspect/(float)a.w*slope_x, py = 2.0f/(float)a.h*slope_y;
        const float part = STRIP_S == 2 ? 0.30f : 0.45f;               // of the pixel's diagonal: 0.25 / 0.375 exactly, with a margin
        reach_uv = part*sqrtf(px*px + py*py);
        reach_agluv = part*sqrtf(4.0f/((float)a.w*(float)a.w) + 4.0f/((float)a.h*(float)a.h));
        if (!(reach_uv < 1.0f) || !(reach_uv > 0.0f)) tier = false;   // (a degenerate camera: NaN or nothing to bound)
    }
    const size_t pixel_entries = tier ? (size_t)frames*((size_t)a.w + a.h)*sizeof(float4) : 0;
    const size_t bars2 = tier ? (size_t)frames*sp.height*2*sizeof(float2) : 0;
    const size_t classes = tier ? (size_t)frames*t.blocks_x*t.blocks_y*8 : 0;
    const size_t needed = entries + blocks + ysteps + pixel_entries + bars2 + classes;
    if (ctx->vis_tables_bytes < needed) {
        hipStreamSynchronize(s);
        hipFree(ctx->vis_tables); ctx->vis_tables = nullptr; ctx->vis_tables_bytes = 0;
        if (hipMalloc(&ctx->vis_tables, needed) != hipSuccess) return fail(SFX_E_HIP, "visualizer tables of %d frames: out of device memory", frames);
        ctx->vis_tables_bytes = needed;
    }
    t.columns = (float4*)ctx->vis_tables;
    t.rows = t.columns + (size_t)frames*a.wr*VIS_ENTRY_QUADS;
    t.block_x = (int4*)(t.rows + (size_t)frames*a.hr*VIS_ENTRY_QUADS);
    t.block_y = t.block_x + (size_t)frames*t.blocks_x;
    t.ysteps = STRIP_S ? (float4*)(t.block_y + (size_t)frames*t.blocks_y) : nullptr;
    char* after = (char*)ctx->vis_tables + entries + blocks + ysteps;
    t.pixel_columns = tier ? (float4*)after : nullptr;
    t.pixel_rows = tier ? t.pixel_columns + (size_t)frames*a.w : nullptr;
    float2* spread = tier ? (float2*)(after + pixel_entries) : nullptr;
    t.bars2 = spread;
    t.pixel_reach_uv = reach_uv; t.pixel_reach_agluv = reach_agluv;
    uint8_t* wave_classes = tier ? (uint8_t*)(after + pixel_entries + bars2) : nullptr;
    t.wave_classes = wave_classes;
    if (tier) {
        const long count = (long)frames*sp.height*2;
        hipLaunchKernelGGL(k_visualizer_bar_spread, dim3((unsigned)((count + 255)/256)), dim3(256), 0, s, a.tape_bars + (long)a.frame0*a.spectrogram_stride,
                           (long)a.spectrogram_stride, frames, sp.height, sp.repeat_y, spread);
    }
    hipLaunchKernelGGL(k_visualizer_axes, dim3((a.wr + 127)/128 + (a.hr + 127)/128, frames), dim3(128), 0, s, a, t);
    if constexpr (STRIP_S >= 2) {
        if (tier) {
            const long tiles = (long)t.blocks_x*t.blocks_y*8;
            hipLaunchKernelGGL((k_visualizer_classify<STRIP_S, WALK, CG>), dim3((unsigned)((tiles + 255)/256), frames), dim3(256), 0, s, a, t, wave_classes);
        }
    }
    if constexpr (STRIP_S != 0) {
        g_last_kernel = "k_visualizer_strip<" + std::to_string(PITCH) + ", " + std::to_string(ROWS) + ", " + std::to_string(STRIP_S) + ", " + std::to_string(WALK) + ", " + std::to_string(WAVES) + ", " + std::to_string(CG) + (HALF ? ", true>" : ", false>");
        hipLaunchKernelGGL((k_visualizer_strip<PITCH, ROWS, STRIP_S, WALK, WAVES, CG, HALF>), dim3(t.blocks_x*t.blocks_y, 1, frames), dim3(512), 0, s, a, t);
    } else {
        g_last_kernel = "k_visualizer_fast<" + std::to_string(PITCH) + ", " + std::to_string(ROWS) + ", 128, " + std::to_string(WAVES) + ">";
        hipLaunchKernelGGL((k_visualizer_fast<PITCH, ROWS, 128, WAVES>), dim3(t.blocks_x*t.blocks_y, 1, frames), dim3(512), 0, s, a, t);
    }
    return 1;
}

// ---- the fast visualizer path (visualizer_fast.hpp) -----------------------------------------------------------------------
// Identity camera, unorm8 bilinear background, 2x or 4x SSAA, the window of a block inside one of the compiled tiles, and a blur
// whose axis lines fit their slots at the largest radius the launch can see. Returns 1 when it launched, 0 when the configuration
// is not its own (the caller then takes VisualizerShader), < 0 on errors.
#ifndef VIS_FAST
#define VIS_FAST 1
#endif
#ifndef VIS_FAST_WALK
#define VIS_FAST_WALK 8
#endif
int launch_visualizer_fast(Context* ctx, const RenderArgs& a0, int ssaa, int frames, hipStream_t s, bool to_screen) {
    // fused: 2x or 4x SSAA into the RGB8 frame; to_screen: the samples themselves into an RGBA8 iScreen (the two-pass configuration)
    if (to_screen ? (ssaa != 1 || a0.out_dtype != DT_U8 || a0.out_components != 4) : (ssaa != 2 && ssaa != 4)) return 0;
    if (!VIS_FAST || !ctx || !(a0.identity_camera || a0.axis_camera) || !visualizer_tile_applicable(a0.tex[TEX_BACKGROUND])) return 0;
    const char* toggle = getenv("SHADERFLOW_VIS_FAST");              // A/B switch for measurements: 0 = round 1's kernels, 1 = k_visualizer_fast
    if (toggle && atoi(toggle) == 0) return 0;
    const Tex& bg = a0.tex[TEX_BACKGROUND];
    const Tex& sp = a0.tex[TEX_SPECTROGRAM];
    RenderArgs a = a0;
    if (!a.tape_bars) {
        // a bound one-column RG32F spectrogram: the bar heights per texel into the context's scratch (k_visualizer_bars)
        if (a.dyn || a.tape_spectrogram || !sp.data || sp.dtype != DT_F32 || sp.components != 2 || sp.width != 1 || sp.filter != FILTER_NEAREST) return 0;
        const size_t count = (size_t)sp.height*2;
        if (ctx->vis_bars_count < count) {
            hipStreamSynchronize(s);
            hipFree(ctx->vis_bars); ctx->vis_bars = nullptr; ctx->vis_bars_count = 0;
            if (hipMalloc(&ctx->vis_bars, sizeof(float)*count) != hipSuccess) return fail(SFX_E_HIP, "visualizer bar table: out of device memory");
            ctx->vis_bars_count = count;
        }
        visualizer_bars((const float*)sp.data, (long)count, ctx->vis_bars, s);
        a.tape_bars = ctx->vis_bars; a.spectrogram_stride = 0;
    } else if (sp.width != 1 || sp.components != 2) return 0;
    // the axis lines must fit their slots at the largest blur radius of the launch (visualizer_window_bound's conventions)
    const float intensity = a.has_vis ? fabsf(a.vis.intensity) : 0.003f;
    const float ax = intensity*a.bg_scale_x*(float)bg.width;
    const float reach_line = (fabsf(a.tap_x[0]) + 9.0f*fabsf(a.tap_x[1] - a.tap_x[0]))*ax;
    if (!(2.0f*reach_line + 1.0e-3f < (float)(VIS_LINE_CELLS - 1))) return 0;
    auto fits = [&](int columns, int rows, int pitch, int tile_rows) {
        int tw = 0, th = 0;
        visualizer_window_bound(a, columns, rows, tw, th);
        return tw <= pitch && th <= tile_rows;
    };
    // build knobs of the strip kernel (tools/variants.sh): rows a lane walks at 2x / 4x SSAA, rows of cells of the tiles, resident waves per SIMD
#ifndef VIS_STRIP_WALK2
#define VIS_STRIP_WALK2 (VIS_FAST_WALK == 8 ? 9 : VIS_FAST_WALK)     // nine rows: the longest strip that stays inside 64 VGPRs (ten spill), +2 % over eight
#endif
#ifndef VIS_STRIP_WALK4
#define VIS_STRIP_WALK4 (VIS_FAST_WALK == 8 ? 10 : VIS_FAST_WALK)
#endif
#ifndef VIS_STRIP_ROWS2
#define VIS_STRIP_ROWS2 (VIS_STRIP_WALK2 <= 4 ? 10 : (VIS_STRIP_WALK2 <= 6 ? 11 : 12))
#endif
#ifndef VIS_STRIP_ROWS4
#define VIS_STRIP_ROWS4 13
#endif
#ifndef VIS_STRIP_PITCH2
#define VIS_STRIP_PITCH2 72
#endif
#ifndef VIS_STRIP_WAVES2
#define VIS_STRIP_WAVES2 6                                             // what the kernel HAS: 51.8 KB of LDS per block = three blocks = six waves per SIMD (78 registers); asking for 8 only made the compiler say so on every build
#endif
#ifndef VIS_STRIP_WAVES4
#define VIS_STRIP_WAVES4 6
#endif
    constexpr int WALK2 = VIS_STRIP_WALK2, WALK4 = VIS_STRIP_WALK4;
    const bool plain = toggle && atoi(toggle) == 1;                   // force the quad-per-pixel kernel
    if (ssaa == 1) {
        // no SSAA: 0.87 texel per sample at 1080p over a 1080-row background — nothing to share along a strip, but the tables, the
        // folded tap pairs and the speculated post-processing still apply. 64 columns x 8 rows per block over a 66 x 15 tile
        // (three blocks per CU), or strips of two rows over 66 x 22 (two)
#ifndef VIS_STRIP_WALK1
#define VIS_STRIP_WALK1 2
#endif
#ifdef VIS_STRIP_ROWS1                                               // (A/B builds: longer strips without SSAA, tools/variants.sh)
        if (fits(64, 8*VIS_STRIP_WALK1, 66, VIS_STRIP_ROWS1)) return launch_visualizer_tables_and_kernel<66, VIS_STRIP_ROWS1, 1, VIS_STRIP_WALK1, VIS_STRIP_WAVES1, 1>(ctx, a, frames, s);
#endif
        if (VIS_STRIP_WALK1 == 2 && fits(64, 16, 66, 22)) return launch_visualizer_tables_and_kernel<66, 22, 1, 2, 4, 1>(ctx, a, frames, s);
        if (fits(64, 8, 66, 15)) return launch_visualizer_tables_and_kernel<66, 15, 1, 1, 6, 1>(ctx, a, frames, s);
    } else if (ssaa == 2) {
#ifndef VIS_STRIP_HALF2
#define VIS_STRIP_HALF2 false
#endif
// float16 cells (visualizer_fast.hpp): where the lanes of a wave read DIFFERENT cells the kernel is bound by LDS bandwidth and
// half the bytes win — 1080p 2x 7 200 -> 7 970, 1440p 2x 4 480 -> 4 920, 720p 2x 9 430 -> 13 270 frames/s (profiles/r03_variants.txt);
// at 4K 2x the lanes share their cells and plain float32 multiply-adds are cheaper
#ifndef VIS_DENSE_HALF
#define VIS_DENSE_HALF true
#endif
#ifndef VIS_SPARSE_HALF
#define VIS_SPARSE_HALF true
#endif
#ifndef VIS_STRIP_HALF4
#define VIS_STRIP_HALF4 false
#endif
#ifndef VIS_MID4_HALF
#define VIS_MID4_HALF false
#endif
#ifndef VIS_SPARSE4_HALF
#define VIS_SPARSE4_HALF false
#endif
        if (VIS_FAST_WALK > 0 && !plain && fits(256, 2*WALK2, VIS_STRIP_PITCH2, VIS_STRIP_ROWS2)) return launch_visualizer_tables_and_kernel<VIS_STRIP_PITCH2, VIS_STRIP_ROWS2, 2, WALK2, VIS_STRIP_WAVES2, 4, VIS_STRIP_HALF2>(ctx, a, frames, s);
        // denser outputs (1080p or 1440p at 2x SSAA over a 1080-row background: up to 0.43 texel per sample): strips of six rows
        // over a 120 x 13 tile, two blocks per CU
#ifndef VIS_DENSE_WALK
#define VIS_DENSE_WALK 6                                              // the longest strip whose 120-cell-wide tile still leaves two blocks per CU
#endif
#ifndef VIS_DENSE_ROWS
#define VIS_DENSE_ROWS 13
#endif
        if (VIS_FAST_WALK > 0 && !plain && fits(256, 2*VIS_DENSE_WALK, 120, VIS_DENSE_ROWS)) return launch_visualizer_tables_and_kernel<120, VIS_DENSE_ROWS, 2, VIS_DENSE_WALK, 4, 4, VIS_DENSE_HALF>(ctx, a, frames, s);
        // sparser still (720p at 2x SSAA: 0.65 texel per sample): 128 columns x 12 rows per block, strips of three, 92 x 16 tile
#ifndef VIS_SPARSE_WALK
#define VIS_SPARSE_WALK 3
#endif
#ifndef VIS_SPARSE_ROWS
#define VIS_SPARSE_ROWS 16
#endif
        if (VIS_FAST_WALK > 0 && !plain && fits(128, 4*VIS_SPARSE_WALK, 92, VIS_SPARSE_ROWS)) return launch_visualizer_tables_and_kernel<92, VIS_SPARSE_ROWS, 2, VIS_SPARSE_WALK, 4, 2, VIS_SPARSE_HALF>(ctx, a, frames, s);
        if (fits(256, 2, 72, 10)) return launch_visualizer_tables_and_kernel<72, 10, 0, 0, 8>(ctx, a, frames, s);
    } else if (VIS_FAST_WALK > 0) {
#ifdef VIS4_SWEEP_WALK                                               // (A/B builds: BASELINE config 4's LDS-tile sweep — tools/experiments/c4_tile_sweep.sh)
        if (fits(64*VIS4_SWEEP_CG, (8/VIS4_SWEEP_CG)*VIS4_SWEEP_WALK, VIS4_SWEEP_PITCH, VIS4_SWEEP_ROWS))
            return launch_visualizer_tables_and_kernel<VIS4_SWEEP_PITCH, VIS4_SWEEP_ROWS, 4, VIS4_SWEEP_WALK, VIS4_SWEEP_WAVES, VIS4_SWEEP_CG, false>(ctx, a, frames, s);
#endif
        // 8K at 4x over a 1080-row background (0.0625 texel per sample): a block's window is 17 cells wide — on a 20-cell pitch its staging loop
        // has no idle half (round 6's tile sweep, profiles/r06_c4_tile_sweep.txt: 292.0 -> 295.9 frames/s, the same bytes); wider windows
        // (smaller outputs, larger backgrounds) keep the 40-cell pitch
        if (fits(128, 4*WALK4, 20, VIS_STRIP_ROWS4)) return launch_visualizer_tables_and_kernel<20, VIS_STRIP_ROWS4, 4, WALK4, VIS_STRIP_WAVES4, 2, VIS_STRIP_HALF4>(ctx, a, frames, s);
        if (fits(128, 4*WALK4, 40, VIS_STRIP_ROWS4)) return launch_visualizer_tables_and_kernel<40, VIS_STRIP_ROWS4, 4, WALK4, VIS_STRIP_WAVES4, 2, VIS_STRIP_HALF4>(ctx, a, frames, s);
        // 1080p at 4x SSAA: the same tile width with strips of six rows
#ifndef VIS_MID4_WALK
#define VIS_MID4_WALK 6
#endif
#ifndef VIS_MID4_ROWS
#define VIS_MID4_ROWS 14
#endif
#ifndef VIS_SPARSE4_WALK
#define VIS_SPARSE4_WALK 6
#endif
#ifndef VIS_SPARSE4_ROWS
#define VIS_SPARSE4_ROWS 16
#endif
        if (WALK4 != VIS_MID4_WALK && fits(128, 4*VIS_MID4_WALK, 40, VIS_MID4_ROWS)) return launch_visualizer_tables_and_kernel<40, VIS_MID4_ROWS, 4, VIS_MID4_WALK, 8, 2, VIS_MID4_HALF>(ctx, a, frames, s);
        // 720p at 4x SSAA (0.32 texel per sample)
        if (fits(128, 4*VIS_SPARSE4_WALK, 56, VIS_SPARSE4_ROWS)) return launch_visualizer_tables_and_kernel<56, VIS_SPARSE4_ROWS, 4, VIS_SPARSE4_WALK, 6, 2, VIS_SPARSE4_HALF>(ctx, a, frames, s);
    }
    return 0;
}

}  // namespace sfl
