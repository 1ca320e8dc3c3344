// launch_visualizer_tiled.hip — visualizer.frag over an LDS tile of background cells, one sample per lane (visualizer_kernels.hpp
// VisualizerShader: round 1's kernels): what runs under rolled / tilted cameras and for windows no table-driven kernel takes. The
// choice between the shapes is capi.hip's (launch_render / launch_fused). One of the launch units of libshaderflow_hip.so (launch.hpp).
#include "launch.hpp"
#include "launch_templates.hpp"
#include "visualizer_kernels.hpp"

using namespace sf;

#ifndef VIS_PITCH_SS
#define VIS_PITCH_SS 80
#endif
#ifndef VIS_FUSED_ROWS
#define VIS_FUSED_ROWS 1
#endif
#ifndef VIS_THREAD_ROWS
#define VIS_THREAD_ROWS 1
#endif
#ifndef VIS_BLOCK_PX
#define VIS_BLOCK_PX 128
#endif
#ifndef VIS_ROWS_SS
#define VIS_ROWS_SS 10
#endif
#ifndef VIS_MIN_WAVES_SS
#define VIS_MIN_WAVES_SS 8
#endif
#ifndef VIS_MIN_WAVES_S4
#define VIS_MIN_WAVES_S4 6
#endif

namespace sfl {

void tiled_fused_limits(int& pitch_ss, int& rows_ss, int& block_px, int& block_rows) {
    pitch_ss = VIS_PITCH_SS; rows_ss = VIS_ROWS_SS; block_px = VIS_BLOCK_PX; block_rows = VIS_FUSED_ROWS*VIS_THREAD_ROWS;
}

int render_visualizer_tiled(TiledRender shape, const RenderArgs& a, int frames, hipStream_t s, size_t dynamic_lds) {
    switch (shape) {
        case TILED_R_64x15_WALK8: launch_render_t<VisualizerShader<64, 15, 6, 1, 1, 128, 64, 8>>(a, frames, s); break;
        case TILED_R_DYNAMIC_WALK8: launch_render_t<VisualizerShader<0, 0, 4, 1, 1, 128, 64, 8>>(a, frames, s, dynamic_lds); break;
        case TILED_R_128x10: launch_render_t<VisualizerShader<128, 10, 1>>(a, frames, s); break;
        case TILED_R_DYNAMIC: launch_render_t<VisualizerShader<0, 0, 1>>(a, frames, s, dynamic_lds); break;
        default: return fail(SFX_E_INVALID, "tiled visualizer render shape %d", (int)shape);
    }
    return SFX_OK;
}

// (each shape is compiled for the supersampling factors capi.hip's launch_fused can send it: see there)
int fused_visualizer_tiled(TiledFused shape, const RenderArgs& a, int ssaa, int frames, hipStream_t s, size_t dynamic_lds) {
    switch (shape) {
        case TILED_F_128x10: return launch_fused_s<VisualizerShader<128, 10, 1>, 1>(a, ssaa, frames, s);
        case TILED_F_SS_S4: return launch_fused_s<VisualizerShader<VIS_PITCH_SS, VIS_ROWS_SS, VIS_MIN_WAVES_S4, VIS_FUSED_ROWS>, 4>(a, ssaa, frames, s);
        case TILED_F_SS: return launch_fused_s<VisualizerShader<VIS_PITCH_SS, VIS_ROWS_SS, VIS_MIN_WAVES_SS, VIS_FUSED_ROWS, VIS_THREAD_ROWS, VIS_BLOCK_PX>, 2>(a, ssaa, frames, s);
        case TILED_F_64x11: return launch_fused_s<VisualizerShader<64, 11, 8, 1, 2, 64>, 2>(a, ssaa, frames, s);
        case TILED_F_56x14: return launch_fused_s<VisualizerShader<56, 14, VIS_MIN_WAVES_S4, 1, 4, 32>, 4>(a, ssaa, frames, s);
        case TILED_F_DYN_128: return launch_fused_s<VisualizerShader<0, 0, 4, VIS_FUSED_ROWS, VIS_THREAD_ROWS, 128>, 2, 4>(a, ssaa, frames, s, dynamic_lds);
        case TILED_F_DYN_64x2: return launch_fused_s<VisualizerShader<0, 0, 8, 1, 2, 64>, 2>(a, ssaa, frames, s, dynamic_lds);
        case TILED_F_DYN_32x4: return launch_fused_s<VisualizerShader<0, 0, 8, 1, 4, 32>, 2>(a, ssaa, frames, s, dynamic_lds);
        case TILED_F_DYN_32x4_WALK4: return launch_fused_s<VisualizerShader<0, 0, 4, 4, 4, 32>, 2>(a, ssaa, frames, s, dynamic_lds);
        case TILED_F_DYN_64: return launch_fused_s<VisualizerShader<0, 0, 4, VIS_FUSED_ROWS, VIS_THREAD_ROWS, 64>, 2, 4>(a, ssaa, frames, s, dynamic_lds);
        case TILED_F_DYN_32: return launch_fused_s<VisualizerShader<0, 0, 4, VIS_FUSED_ROWS, VIS_THREAD_ROWS, 32>, 2, 4>(a, ssaa, frames, s, dynamic_lds);
        case TILED_F_DYN_1X: return launch_fused_s<VisualizerShader<0, 0, 4>, 1>(a, ssaa, frames, s, dynamic_lds);
        default: return fail(SFX_E_INVALID, "tiled visualizer fused shape %d", (int)shape);
    }
}

}  // namespace sfl
