// launch_separable.hip — bars.frag / waveform.frag / default.glsl on per-frame column and row tables (separable_fast.hpp) and the second
// layers of multipass.frag / motionblur.frag (layered_fast.hpp). One of the launch units of libshaderflow_hip.so (launch.hpp).
#include "launch.hpp"
#include "separable_fast.hpp"
#include "layered_fast.hpp"

#include <cstdlib>

using namespace sf;

namespace sfl {

// ---- the second layer of multipass.frag / motionblur.frag (layered_fast.hpp): 1 launched, 0 not this path's configuration -----------
#ifndef LAYERED_FAST
#define LAYERED_FAST 1
#endif
int launch_multipass_layer1(Context* ctx, const RenderArgs& a, int frames, hipStream_t s) {
    const Tex& first = a.tex[TEX_HISTORY];
    static const bool off = [] { const char* e = getenv("SHADERFLOW_LAYERED_FAST"); return e && atoi(e) == 0; }();     // A/B switch for measurements
    if (!LAYERED_FAST || off || !ctx || a.u.iLayer != 1 || !first.data || first.dtype != DT_U8 || first.components != 4 || first.filter != FILTER_LINEAR) return 0;
    if (!ctx->multipass_taps) {
        MultipassTaps table;
        multipass_tap_table(table, 5.0f, 8, 8);                     // multipass.frag:41 blur(iScreen0x0, astuv, 5, 8, 8)
        if (table.count > LAYERED_MAX_TAPS) return 0;
        if (hipMalloc((void**)&ctx->multipass_taps, sizeof table) != hipSuccess) return fail(SFX_E_HIP, "multipass tap table: out of device memory");
        if (hipMemcpy(ctx->multipass_taps, &table, sizeof table, hipMemcpyHostToDevice) != hipSuccess) return fail(SFX_E_HIP, "multipass tap table: upload failed");
        ctx->multipass_reach[0] = table.reach_u; ctx->multipass_reach[1] = table.reach_v;
    }
    // texels under a block of 64 x 8 pixels plus the blur's reach on both sides (and the bilinear neighbour, and a texel of slack per side)
    const int tile_w = (int)ceilf((float)MP_BLOCK_W*(float)first.width/(float)a.wr + 2.0f*ctx->multipass_reach[0]*(float)first.width) + 6;
    const int tile_h = (int)ceilf((float)MP_BLOCK_H*(float)first.height/(float)a.hr + 2.0f*ctx->multipass_reach[1]*(float)first.height) + 6;
    const size_t lds = (size_t)tile_w*tile_h*sizeof(float4);
    if (lds > 96*1024) return 0;                                    // a layer far larger than its target: the generic kernel
    if (lds > 48*1024) hipFuncSetAttribute((const void*)k_multipass_layer1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    g_last_kernel = "k_multipass_layer1";
    hipLaunchKernelGGL(k_multipass_layer1, dim3((a.wr + MP_BLOCK_W - 1)/MP_BLOCK_W, (a.hr + MP_BLOCK_H - 1)/MP_BLOCK_H, frames), dim3(MP_BLOCK_W, MP_BLOCK_H/MP_ROWS), lds, s,
                       a, ctx->multipass_taps, tile_w, tile_h);
    return 1;
}
int launch_motionblur_layer1(Context*, const RenderArgs& a, int frames, hipStream_t s) {
    static const bool off = [] { const char* e = getenv("SHADERFLOW_LAYERED_FAST"); return e && atoi(e) == 0; }();
    const int temporal = (int)a.u.user[USER_SCREEN_TEMPORAL];
    if (!LAYERED_FAST || off || a.u.iLayer != 1 || temporal < 1 || temporal > TEX_HISTORY_DEPTH) return 0;
    const Tex& first = a.tex[TEX_HISTORY];
    MotionblurArgs m{};
    m.temporal = temporal;
    for (int t = 0; t < temporal; t++) {
        const Tex& layer = a.tex[TEX_HISTORY + t];
        // one size, one sampler state: then addressing and weights of texture(iScreen{t}x0, astuv) are one computation per pixel
        if (!layer.data || layer.dtype != DT_U8 || layer.components != 4 || layer.filter != FILTER_LINEAR || layer.width != first.width || layer.height != first.height
            || layer.repeat_x != first.repeat_x || layer.repeat_y != first.repeat_y) return 0;
        m.layer[t] = (const uint32_t*)layer.data;
        m.factor[t] = sf::smoothstep(1.0f, 0.0f, (float)t/(float)temporal);                  // motionblur.frag:11
    }
    g_last_kernel = "k_motionblur_layer1";
    hipLaunchKernelGGL(k_motionblur_layer1, dim3((a.wr + 63)/64, (a.hr + 3)/4, frames), dim3(64, 4), 0, s, a, m);
    return 1;
}

// ---- bars.frag / waveform.frag with per-frame column and row tables (separable_fast.hpp) ---------------------------------------
// 1 launched, 0 not this path's configuration (the caller takes PlainShader), < 0 error.
template <int KIND> static int launch_separable_t(Context* ctx, const RenderArgs& a, int ssaa, int frames, hipStream_t s) {
    if (!ctx || ssaa != 2) return 0;
    if (getenv("SHADERFLOW_SEPARABLE") && atoi(getenv("SHADERFLOW_SEPARABLE")) == 0) return 0;        // A/B switch for measurements
    if (KIND == SEP_DEFAULT && !(a.identity_camera || a.axis_camera)) return 0;   // default.glsl reads iCamera.gluv: separable per axis without a rotation only
    if (KIND == SEP_BARS) {
        // a one-column spectrogram picked with nearest filtering: the look-up is a function of the sample column alone
        const Tex& sp = a.tex[TEX_SPECTROGRAM];
        if (sp.width != 1 || sp.filter != FILTER_NEAREST) return 0;
    }
    if (KIND == SEP_WAVEFORM && a.hr > 0xffff) return 0;              // its column entries pack a first row and a row count into 16 bits each
    // default.glsl in two passes (separable_fast.hpp k_default_quads): a byte per group of four rows and block of 256 pixels, after the tables
    // OPT-IN (SHADERFLOW_DEFAULT_QUADS=1, read per launch so that a test can turn it on): measured slower than the one pass — a smooth
    // frame takes 6.3 us either way, and the second pass pays for skipping (profiles/r05_basic_quads.txt)
    const char* quads_env = getenv("SHADERFLOW_DEFAULT_QUADS");
    const bool quads = KIND == SEP_DEFAULT && quads_env && atoi(quads_env) == 1 && (a.w & 3) == 0 && a.w >= 256 && a.h >= 4;
    const int done_groups = (a.h + 3)/4, done_blocks = (a.w + 255)/256;
    const size_t tables = (size_t)frames*((size_t)a.wr + a.hr)*sizeof(float4);
    const size_t bytes = tables + (quads ? (size_t)frames*done_groups*done_blocks : 0);
    if (ctx->vis_tables_bytes < bytes) {
        hipStreamSynchronize(s);
        hipFree(ctx->vis_tables); ctx->vis_tables = nullptr; ctx->vis_tables_bytes = 0;
        if (hipMalloc(&ctx->vis_tables, bytes) != hipSuccess) return fail(SFX_E_HIP, "column/row tables of %d frames: out of device memory", frames);
        ctx->vis_tables_bytes = bytes;
    }
    SepTables t;
    t.columns = (float4*)ctx->vis_tables;
    t.rows = t.columns + (size_t)frames*a.wr;
    t.done = quads ? (uint8_t*)ctx->vis_tables + tables : nullptr;
    t.done_groups = done_groups; t.done_blocks = done_blocks;
    hipLaunchKernelGGL(k_separable_axis<KIND>, dim3((a.wr + a.hr + 255)/256, frames), dim3(256), 0, s, a, t);
    if constexpr (KIND != SEP_DEFAULT) {
        // rows as runs, four pixels = one 12-byte store per lane (k_separable_runs); odd widths keep the per-pixel kernel
        const char* runs = getenv("SHADERFLOW_SEPARABLE_RUNS");     // A/B switch for measurements
        if (a.w % 4 == 0 && !(runs && atoi(runs) == 0)) {
            g_last_kernel = std::string("k_separable_runs<") + (KIND == SEP_BARS ? "bars" : "waveform") + ">";
            hipLaunchKernelGGL(k_separable_runs<KIND>, dim3((a.w + 255)/256, (a.h + 4*SEP_RUN_ROWS - 1)/(4*SEP_RUN_ROWS), frames), dim3(256), 0, s, a, t);
            return 1;
        }
    }
    g_last_kernel = std::string("k_separable_fused<") + (KIND == SEP_BARS ? "bars" : (KIND == SEP_WAVEFORM ? "waveform" : "default")) + ">";
    const int blocks_x = (a.w + SEP_PIXELS - 1)/SEP_PIXELS;
    // default.glsl's singular line (separable_fast.hpp k_default_singular), behind whichever kernels rendered the frames
    struct SingularLine { const RenderArgs& a; int frames; hipStream_t s; bool armed;
                          ~SingularLine() { if (armed) hipLaunchKernelGGL(k_default_singular, dim3((a.wr + a.hr + 255)/256, frames), dim3(256), 0, s, a); } } singular{a, frames, s, KIND == SEP_DEFAULT};
    if constexpr (KIND == SEP_DEFAULT) {
        // the smooth tier first, four pixels per lane; what it writes it marks, and the second pass skips
        if (quads) {
            hipLaunchKernelGGL(k_default_quads, dim3(done_blocks, (a.h + 4*DQ_ROWS*DQ_WALKS - 1)/(4*DQ_ROWS*DQ_WALKS), frames), dim3(256), 0, s, a, t);
            hipLaunchKernelGGL((k_separable_fused<KIND, SEP_ROWS_DEFAULT, 1, true>), dim3(blocks_x, (a.h + SEP_ROWS_DEFAULT - 1)/SEP_ROWS_DEFAULT, frames), dim3(SEP_PIXELS), 0, s, a, t);
            return 1;
        }
    }
    if (KIND == SEP_DEFAULT && (long)blocks_x*((a.h + SEP_ROWS_DEFAULT*SEP_CHUNKS_DEFAULT - 1)/(SEP_ROWS_DEFAULT*SEP_CHUNKS_DEFAULT))*frames >= 8192)
        hipLaunchKernelGGL((k_separable_fused<KIND, SEP_ROWS_DEFAULT, SEP_CHUNKS_DEFAULT>), dim3(blocks_x, (a.h + SEP_ROWS_DEFAULT*SEP_CHUNKS_DEFAULT - 1)/(SEP_ROWS_DEFAULT*SEP_CHUNKS_DEFAULT), frames), dim3(SEP_PIXELS), 0, s, a, t);
    else if (KIND == SEP_DEFAULT && (long)blocks_x*((a.h + SEP_ROWS_DEFAULT - 1)/SEP_ROWS_DEFAULT)*frames >= 2048)
        hipLaunchKernelGGL((k_separable_fused<KIND, SEP_ROWS_DEFAULT>), dim3(blocks_x, (a.h + SEP_ROWS_DEFAULT - 1)/SEP_ROWS_DEFAULT, frames), dim3(SEP_PIXELS), 0, s, a, t);
    else if ((long)blocks_x*((a.h + SEP_ROWS_LARGE - 1)/SEP_ROWS_LARGE)*frames >= 2048)
        hipLaunchKernelGGL((k_separable_fused<KIND, SEP_ROWS_LARGE>), dim3(blocks_x, (a.h + SEP_ROWS_LARGE - 1)/SEP_ROWS_LARGE, frames), dim3(SEP_PIXELS), 0, s, a, t);
    else
        hipLaunchKernelGGL((k_separable_fused<KIND, SEP_ROWS_SMALL>), dim3(blocks_x, (a.h + SEP_ROWS_SMALL - 1)/SEP_ROWS_SMALL, frames), dim3(SEP_PIXELS), 0, s, a, t);
    return 1;
}

static_assert(SEPARABLE_BARS == SEP_BARS && SEPARABLE_WAVEFORM == SEP_WAVEFORM && SEPARABLE_DEFAULT == SEP_DEFAULT, "launch.hpp names separable_fast.hpp's kinds");
int launch_separable(int kind, Context* ctx, const RenderArgs& a, int ssaa, int frames, hipStream_t s) {
    if (kind == SEP_DEFAULT) return launch_separable_t<SEP_DEFAULT>(ctx, a, ssaa, frames, s);
    if (kind == SEP_BARS) return launch_separable_t<SEP_BARS>(ctx, a, ssaa, frames, s);
    if (kind == SEP_WAVEFORM) return launch_separable_t<SEP_WAVEFORM>(ctx, a, ssaa, frames, s);
    return 0;
}

}  // namespace sfl
