// launch.hpp — the launch units of libshaderflow_hip.so. Every kernel family is instantiated in ONE translation unit (a clean build runs
// them side by side under `make -j`, an edit of a kernel header rebuilds its unit only); capi.hip (with capi_readout.hip and capi_audio.hip) keeps the C-ABI, the objects and the
// choice between the families, and calls the units through these functions. Return conventions as before the split: the `launch_*`
// pickers return 1 when they launched, 0 when the configuration is not theirs (the caller takes the next family), < 0 on errors.
#pragma once

#include "host_state.hpp"

namespace sfl {

// launch_generic.hip — PlainShader<fragment>: k_render / k_render_resolve of every registered fragment
int render_plain(int fragment, const sf::RenderArgs& a, int frames, hipStream_t s);
int fused_plain(int fragment, const sf::RenderArgs& a, int ssaa, int frames, hipStream_t s);

// launch_visualizer_tiled.hip — VisualizerShader<…> over an LDS tile (round 1's kernels: rolled / tilted cameras, windows no table kernel takes)
enum TiledRender { TILED_R_64x15_WALK8, TILED_R_DYNAMIC_WALK8, TILED_R_128x10, TILED_R_DYNAMIC };
enum TiledFused { TILED_F_128x10, TILED_F_SS_S4, TILED_F_SS, TILED_F_64x11, TILED_F_56x14, TILED_F_DYN_128, TILED_F_DYN_64x2, TILED_F_DYN_32x4, TILED_F_DYN_32x4_WALK4,
                  TILED_F_DYN_64, TILED_F_DYN_32, TILED_F_DYN_1X };
void tiled_fused_limits(int& pitch_ss, int& rows_ss, int& block_px, int& block_rows);      // the build's fixed 2x / 4x tile and block
int render_visualizer_tiled(TiledRender shape, const sf::RenderArgs& a, int frames, hipStream_t s, size_t dynamic_lds);
int fused_visualizer_tiled(TiledFused shape, const sf::RenderArgs& a, int ssaa, int frames, hipStream_t s, size_t dynamic_lds);

// launch_visualizer_strip.hip — k_visualizer_axes + k_visualizer_strip<…> / k_visualizer_fast<…> (visualizer_fast.hpp)
void visualizer_consts_frames(const sf::FrameDyn* dyn, int frame0, int nframes, sf::VisualizerConsts* out, hipStream_t s);
void visualizer_bars(const float* columns, long count, float* bars, hipStream_t s);
int launch_visualizer_fast(Context* ctx, const sf::RenderArgs& a, int ssaa, int frames, hipStream_t s, bool to_screen);

// launch_separable.hip — bars / waveform / default.glsl on per-frame column and row tables (separable_fast.hpp), the second layers of
// multipass.frag / motionblur.frag (layered_fast.hpp)
enum SeparableKind { SEPARABLE_BARS = 0, SEPARABLE_WAVEFORM = 1, SEPARABLE_DEFAULT = 2 };      // = separable_fast.hpp's SEP_* (checked there)
int launch_separable(int kind, Context* ctx, const sf::RenderArgs& a, int ssaa, int frames, hipStream_t s);
int launch_multipass_layer1(Context* ctx, const sf::RenderArgs& a, int frames, hipStream_t s);
int launch_motionblur_layer1(Context* ctx, const sf::RenderArgs& a, int frames, hipStream_t s);

// launch_resolve.hip — final.glsl as a pass (resolve_fast.hpp, render_kernels.hpp k_resolve)
int launch_resolve(Context* ctx, const sf::ResolveArgs& a, int frames, hipStream_t s);

}  // namespace sfl
