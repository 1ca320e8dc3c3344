// launch_resolve.hip — final.glsl as a pass of its own (shader.py:391-396): resolve_fast.hpp's table-driven kernels, else the generic
// k_resolve. One of the launch units of libshaderflow_hip.so (launch.hpp).
#define SF_UNIT_RESOLVE
#include "launch.hpp"
#include "resolve_fast.hpp"

#include <cstdlib>

using namespace sf;

namespace sfl {

// final.glsl as a pass: k_resolve_fast (resolve_fast.hpp) for a linear, clamped iScreen and the kernels it is compiled for, else
// the generic k_resolve. `frames` launches share the tables (they depend on the sizes only).
#ifndef RESOLVE_FAST
#define RESOLVE_FAST 1
#endif
int launch_resolve(Context* ctx, const ResolveArgs& a, int frames, hipStream_t s) {
    const char* toggle = getenv("SHADERFLOW_RESOLVE_FAST");         // A/B switch for measurements
    const bool fast = RESOLVE_FAST && ctx && !(toggle && atoi(toggle) == 0) && a.screen.filter == FILTER_LINEAR && !a.screen.repeat_x && !a.screen.repeat_y &&
                      a.subsample >= 1 && a.subsample <= 3 && a.screen.width > 0 && a.screen.height > 0;
    if (fast) {
        const size_t bytes = ((size_t)a.w + a.h)*a.subsample*sizeof(int4);
        if (ctx->resolve_tables_bytes < bytes) {
            hipStreamSynchronize(s);
            hipFree(ctx->resolve_tables); ctx->resolve_tables = nullptr; ctx->resolve_tables_bytes = 0; memset(ctx->resolve_tables_key, 0, sizeof ctx->resolve_tables_key);
            if (hipMalloc(&ctx->resolve_tables, bytes) != hipSuccess) return fail(SFX_E_HIP, "resolve tables: out of device memory");
            ctx->resolve_tables_bytes = bytes;
        }
        int4* columns = (int4*)ctx->resolve_tables; int4* rows = columns + (size_t)a.w*a.subsample;
        // the tables are a function of the geometry alone: a frame loop resolves the same geometry every frame (two launches of
        // ≈ 5 us each per frame saved; a stream other than the one that built them rebuilds)
        const long key[8] = {a.w, a.h, a.screen.width, a.screen.height, a.screen.repeat_x, a.screen.repeat_y, a.subsample, (long)(uintptr_t)s};
        if (memcmp(key, ctx->resolve_tables_key, sizeof key) != 0) {
            hipLaunchKernelGGL(k_resolve_axis<0>, dim3((a.w + 127)/128), dim3(128), 0, s, a, columns);
            hipLaunchKernelGGL(k_resolve_axis<1>, dim3((a.h + 127)/128), dim3(128), 0, s, a, rows);
            memcpy(ctx->resolve_tables_key, key, sizeof key);
        }
        const ResolveTables t{columns, rows};
        const char* tent = getenv("SHADERFLOW_RESOLVE_TENT");        // A/B switch for measurements
        if (a.subsample == 2 && a.screen.width == a.w && a.screen.height == a.h && !(tent && atoi(tent) == 0)) {
            // the two-pass configuration (no SSAA, final.glsl's 3 x 3 tent): four pixels per thread, each texel read once
            hipLaunchKernelGGL(k_resolve_tent, dim3((a.w + TENT_BW - 1)/TENT_BW, (a.h + TENT_BH - 1)/TENT_BH, frames), dim3(TENT_BW, TENT_BH/TENT_ROWS_PER_THREAD), 0, s, a, t);
            return SFX_OK;
        }
        const dim3 grid((a.w + 63)/64, (a.h + 3)/4, frames), block(64, 4);
        // iScreen texels under a block of 64 x 4 pixels, two more per axis for the bilinear neighbours: the LDS window (a block
        // whose own window is larger — it cannot be — or a launch over the cap reads iScreen directly)
        const long tw = ((long)64*a.screen.width + a.w - 1)/a.w + 3, th = ((long)4*a.screen.height + a.h - 1)/a.h + 3;
        const int window = tw*th <= RESOLVE_WINDOW_TEXELS ? (int)(tw*th) : 0;
        const size_t lds = (size_t)window*sizeof(float4);
        if (a.subsample == 1) hipLaunchKernelGGL(k_resolve_fast<1>, grid, block, lds, s, a, t, window);
        else if (a.subsample == 2) hipLaunchKernelGGL(k_resolve_fast<2>, grid, block, lds, s, a, t, window);
        else hipLaunchKernelGGL(k_resolve_fast<3>, grid, block, lds, s, a, t, window);
        return SFX_OK;
    }
    hipLaunchKernelGGL(k_resolve, dim3((a.w + 63)/64, (a.h + 3)/4, frames), dim3(64, 4), 0, s, a);
    return SFX_OK;
}

}  // namespace sfl
