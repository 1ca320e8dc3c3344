// launch_templates.hpp — the launch of one instance of the generic kernels (render_kernels.hpp) and the name it leaves in sfx_last_kernel
#pragma once

#include "host_state.hpp"

using namespace sf;

// "k_render<…>" / "k_render_resolve<…, S>" of the instance a launch picked, from the compiler's spelling of the enclosing template
static void note_kernel(const char* pretty, const char* kernel, int ssaa = 0) {
    std::string text(pretty);
    const size_t at = text.find("SHADER = ");
    std::string shader = at == std::string::npos ? text : text.substr(at + 9);
    size_t end = shader.find(", S = ");                             // "[SHADER = …, S = 2]" / "[SHADER = …]"
    if (end == std::string::npos) end = shader.find_first_of(";]");
    if (end != std::string::npos) shader.resize(end);
    for (size_t k; (k = shader.find("sf::")) != std::string::npos; ) shader.erase(k, 4);
    g_last_kernel = std::string(kernel) + "<" + shader + (ssaa ? ", " + std::to_string(ssaa) : std::string()) + ">";
}

template <class SHADER> static void launch_render_t(const RenderArgs& a, int frames, hipStream_t s, size_t dynamic_lds = 0) {
    note_kernel(__PRETTY_FUNCTION__, "k_render");
    dim3 grid((a.wr + SHADER::BLOCK_W - 1)/SHADER::BLOCK_W, (a.hr + SHADER::BLOCK_H - 1)/SHADER::BLOCK_H, frames), block(SHADER::BLOCK_W, SHADER::BLOCK_H, 1);
    if (dynamic_lds > 48*1024) hipFuncSetAttribute((const void*)k_render<SHADER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynamic_lds);
    hipLaunchKernelGGL(k_render<SHADER>, grid, block, dynamic_lds, s, a);
}

template <class SHADER, int S> static void launch_fused_k(const RenderArgs& a, dim3 grid, dim3 block, size_t dynamic_lds, hipStream_t s) {
    note_kernel(__PRETTY_FUNCTION__, "k_render_resolve", S);
    if (dynamic_lds > 48*1024) hipFuncSetAttribute((const void*)k_render_resolve<SHADER, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynamic_lds);
    hipLaunchKernelGGL((k_render_resolve<SHADER, S>), grid, block, dynamic_lds, s, a);
}

// `ALLOWED`: the supersampling factors this shape is ever launched with — only those instances are compiled
template <class SHADER, int... ALLOWED> static int launch_fused_s(const sf::RenderArgs& a, int ssaa, int frames, hipStream_t s, size_t dynamic_lds = 0) {
    using namespace sf;
    constexpr int rows = SHADER::FUSED_ROWS*SHADER::THREAD_ROWS, threads = 4*SHADER::BLOCK_PX*SHADER::THREAD_ROWS;
    const int blocks_x = (a.w + SHADER::BLOCK_PX - 1)/SHADER::BLOCK_PX;          // S >= 2; S == 1 always covers 128 x 2 pixels
    const int row_blocks = (a.h + rows - 1)/rows;
    constexpr bool any = sizeof...(ALLOWED) == 0;
    constexpr bool s1 = any || ((ALLOWED == 1) || ...), s2 = any || ((ALLOWED == 2) || ...), s4 = any || ((ALLOWED == 4) || ...);
    if constexpr (s1) if (ssaa == 1) { launch_fused_k<SHADER, 1>(a, dim3(((a.w + 127)/128)*((a.h + 1)/2), 1, frames), dim3(256), dynamic_lds, s); return SFX_OK; }
    if constexpr (s2) if (ssaa == 2) { launch_fused_k<SHADER, 2>(a, dim3(blocks_x*row_blocks, 1, frames), dim3(threads), dynamic_lds, s); return SFX_OK; }
    if constexpr (s4) if (ssaa == 4) { launch_fused_k<SHADER, 4>(a, dim3(blocks_x*row_blocks, 1, frames), dim3(threads), dynamic_lds, s); return SFX_OK; }
    return fail(SFX_E_UNSUPPORTED, "fused ssaa %d", ssaa);
}
