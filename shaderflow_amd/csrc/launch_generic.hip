// launch_generic.hip — the generic kernels (render_kernels.hpp k_render / k_render_resolve) over PlainShader<fragment>: every fragment the
// reference ships, restated in fragments.hpp, one sample per lane. One of the launch units of libshaderflow_hip.so (launch.hpp).
#include "launch.hpp"
#include "launch_templates.hpp"
#include "visualizer_kernels.hpp"

using namespace sf;

namespace sfl {

int render_plain(int fragment, const RenderArgs& a, int frames, hipStream_t s) {
    switch (fragment) {
        case FRAG_DEFAULT: launch_render_t<PlainShader<FRAG_DEFAULT>>(a, frames, s); break;
        case FRAG_MISSING: launch_render_t<PlainShader<FRAG_MISSING>>(a, frames, s); break;
        case FRAG_VISUALIZER: launch_render_t<PlainShader<FRAG_VISUALIZER>>(a, frames, s); break;
        case FRAG_BARS: launch_render_t<PlainShader<FRAG_BARS>>(a, frames, s); break;
        case FRAG_WAVEFORM: launch_render_t<PlainShader<FRAG_WAVEFORM>>(a, frames, s); break;
        case FRAG_MULTI_CHILD: launch_render_t<PlainShader<FRAG_MULTI_CHILD>>(a, frames, s); break;
        case FRAG_MULTI_MAIN: launch_render_t<PlainShader<FRAG_MULTI_MAIN>>(a, frames, s); break;
        case FRAG_SHADERTOY: launch_render_t<PlainShader<FRAG_SHADERTOY>>(a, frames, s); break;
        case FRAG_DYNAMICS: launch_render_t<PlainShader<FRAG_DYNAMICS>>(a, frames, s); break;
        case FRAG_AUDIO: launch_render_t<PlainShader<FRAG_AUDIO>>(a, frames, s); break;
        case FRAG_MULTIPASS: launch_render_t<PlainShader<FRAG_MULTIPASS>>(a, frames, s); break;
        case FRAG_MOTIONBLUR: launch_render_t<PlainShader<FRAG_MOTIONBLUR>>(a, frames, s); break;
        case FRAG_LIFE_SIMULATION: launch_render_t<PlainShader<FRAG_LIFE_SIMULATION>>(a, frames, s); break;
        case FRAG_LIFE_VISUALS: launch_render_t<PlainShader<FRAG_LIFE_VISUALS>>(a, frames, s); break;
        case FRAG_VIDEO: launch_render_t<PlainShader<FRAG_VIDEO>>(a, frames, s); break;
        case FRAG_RAYMARCH: launch_render_t<PlainShader<FRAG_RAYMARCH>>(a, frames, s); break;
        case FRAG_MANDELBROT: launch_render_t<PlainShader<FRAG_MANDELBROT>>(a, frames, s); break;
        case FRAG_TETRATION: launch_render_t<PlainShader<FRAG_TETRATION>>(a, frames, s); break;
        default: return fail(SFX_E_UNSUPPORTED, "fragment %d has no render kernel", fragment);
    }
    return SFX_OK;
}

int fused_plain(int fragment, const RenderArgs& a, int ssaa, int frames, hipStream_t s) {
    switch (fragment) {
        case FRAG_DEFAULT: return launch_fused_s<PlainShader<FRAG_DEFAULT>>(a, ssaa, frames, s);
        case FRAG_MISSING: return launch_fused_s<PlainShader<FRAG_MISSING>>(a, ssaa, frames, s);
        case FRAG_VISUALIZER: return launch_fused_s<PlainShader<FRAG_VISUALIZER>>(a, ssaa, frames, s);
        case FRAG_BARS: return launch_fused_s<PlainShader<FRAG_BARS>>(a, ssaa, frames, s);
        case FRAG_WAVEFORM: return launch_fused_s<PlainShader<FRAG_WAVEFORM>>(a, ssaa, frames, s);
        case FRAG_MULTI_CHILD: return launch_fused_s<PlainShader<FRAG_MULTI_CHILD>>(a, ssaa, frames, s);
        case FRAG_MULTI_MAIN: return launch_fused_s<PlainShader<FRAG_MULTI_MAIN>>(a, ssaa, frames, s);
        case FRAG_SHADERTOY: return launch_fused_s<PlainShader<FRAG_SHADERTOY>>(a, ssaa, frames, s);
        case FRAG_DYNAMICS: return launch_fused_s<PlainShader<FRAG_DYNAMICS>>(a, ssaa, frames, s);
        case FRAG_AUDIO: return launch_fused_s<PlainShader<FRAG_AUDIO>>(a, ssaa, frames, s);
        case FRAG_LIFE_VISUALS: return launch_fused_s<PlainShader<FRAG_LIFE_VISUALS>>(a, ssaa, frames, s);
        case FRAG_VIDEO: return launch_fused_s<PlainShader<FRAG_VIDEO>>(a, ssaa, frames, s);
        case FRAG_RAYMARCH: return launch_fused_s<PlainShader<FRAG_RAYMARCH>>(a, ssaa, frames, s);
        case FRAG_MANDELBROT: return launch_fused_s<PlainShader<FRAG_MANDELBROT>>(a, ssaa, frames, s);
        case FRAG_TETRATION: return launch_fused_s<PlainShader<FRAG_TETRATION>>(a, ssaa, frames, s);
        default: return fail(SFX_E_UNSUPPORTED, "fragment %d has no fused kernel", fragment);
    }
}

}  // namespace sfl
