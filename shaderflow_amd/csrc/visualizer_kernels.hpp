// visualizer_kernels.hpp — visualizer.frag with its radial blur (examples/basic/shaders/visualizer.frag:21-33)
// evaluated from an LDS-staged tile of the background texture.
//
// The blur is 1 + 9x10 bilinear taps of the background per supersample: ~3.0 G taps per 4K 2xSSAA
// frame, so this loop IS the frame time (FP32 VALU bound, not HBM bound). Every tap of
// every supersample of a block falls inside one small window of the background, so the block:
//   1. reduces the bounding box of its centre taps (wave shuffles + LDS),
//   2. stages that window once — wrapped/clamped per texture.py:274-283 — as float16 texel PAIRS
//      {T[x,y], T[x+1,y]} (16 B per position: one ds_read_b128 per bilinear row),
//   3. runs the taps in tile-local texel coordinates: 2 fma for the position, floor/fract, 5 weight ops,
//      12 v_fma_mix_f32 (f16 texel x f32 weight, f32 accumulate — byte values are exact in f16).
// Direction 8 of the float-counter loop coincides with direction 0 to 3e-9 texel (SURVEY.md §7 hard part
// 4), so direction 0 is evaluated once and counted twice: 81 taps instead of 91.
//
// Accuracy contract: the tap POSITIONS differ from the generic chain (glsl.hpp stexture) by the rounding
// of an affine re-association (≤ 1e-4 texel) and the sum is scaled by 1/255 once instead of per texel:
// ≤ 1e-6 relative on the blurred colour, i.e. far inside the 1 LSB pixel tolerance. Everything outside
// the blur (visualizer_pre / visualizer_post) is the same code as the generic kernel. If the window
// does not fit the tile (huge background, extreme zoom) the block falls back to the generic taps.
#pragma once

#include "render_kernels.hpp"

namespace sf {

constexpr int TILE_PITCH = 128;      // positions per tile row (16 B each)
constexpr int TILE_ROWS = 16;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct VisualizerShader {
    struct State {
        VisualizerPre pre;
        float xc, yc;                // centre tap in texel space (u*w - 0.5, v*h - 0.5)
    };
    struct Shared {
        half8 tile[TILE_ROWS*TILE_PITCH];
        float red[5][8];
        int x0, y0, ok;
    };

    __device__ static void pre(const RenderArgs&, const Frag& f, bool valid, State& s) {
        s.pre = visualizer_pre(f);
        const Tex& bg = f.tex[TEX_BACKGROUND];
        // same chain as stexture() → texture() for the centre tap (glsl.hpp)
        vec2 scale = {(float)bg.height/(float)bg.width, 1.0f};
        vec2 st = gluv2stuv(stuv2gluv(s.pre.bg)*scale);
        s.xc = st.x*(float)bg.width - 0.5f;
        s.yc = st.y*(float)bg.height - 0.5f;
        (void)valid;
    }

    __device__ static float wave_min(float v) {
        for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m));
        return v;
    }

    template <int N>
    __device__ static void setup(const RenderArgs& a, const Tex* tex, State (&s)[N], const bool (&valid)[N], Shared& sh) {
        const Tex& bg = tex[TEX_BACKGROUND];
        const int tid = threadIdx.y*blockDim.x + threadIdx.x, nthreads = blockDim.x*blockDim.y;
        const int wave = tid >> 6, nwaves = (nthreads + 63) >> 6;
        // 1. bounding box of the centre taps (as minima of x, -x, y, -y)
        float lo_x = INFINITY, hi_x = INFINITY, lo_y = INFINITY, hi_y = INFINITY, neg_i = INFINITY;
        bool any = false, bad = false;
#pragma unroll
        for (int n = 0; n < N; n++) {
            if (valid[n] && !s[n].pre.out_of_bounds) {
                any = true;
                bad = bad || !(fabsf(s[n].xc) < 1e8f) || !(fabsf(s[n].yc) < 1e8f) || !(s[n].pre.intensity == s[n].pre.intensity);
                lo_x = fminf(lo_x, s[n].xc); hi_x = fminf(hi_x, -s[n].xc);
                lo_y = fminf(lo_y, s[n].yc); hi_y = fminf(hi_y, -s[n].yc);
                neg_i = fminf(neg_i, -fabsf(s[n].pre.intensity));
            }
        }
        lo_x = wave_min(lo_x); hi_x = wave_min(hi_x); lo_y = wave_min(lo_y); hi_y = wave_min(hi_y); neg_i = wave_min(neg_i);
        const int n_bad = __syncthreads_count(bad ? 1 : 0);
        if ((tid & 63) == 0) { sh.red[0][wave] = lo_x; sh.red[1][wave] = hi_x; sh.red[2][wave] = lo_y; sh.red[3][wave] = hi_y; sh.red[4][wave] = neg_i; }
        __syncthreads();
        if (tid == 0) {
            float m[5];
            for (int k = 0; k < 5; k++) { m[k] = sh.red[k][0]; for (int w = 1; w < nwaves; w++) m[k] = fminf(m[k], sh.red[k][w]); }
            // tap radius in texels: |cos|,|sin| <= 1, walk <= 1.0000001 (visualizer.frag:26-28)
            float intensity = -m[4];                         // uniform over the frame (visualizer.frag:22)
            float rx = fabsf(intensity*((float)bg.height/(float)bg.width)*(float)bg.width)*1.001f + 0.001f;
            float ry = fabsf(intensity*(float)bg.height)*1.001f + 0.001f;
            int ok = 0, x0 = 0, y0 = 0;
            if (n_bad == 0 && m[0] < INFINITY && rx == rx && rx < 64.0f && ry < 64.0f) {
                x0 = (int)floorf(m[0] - rx); y0 = (int)floorf(m[2] - ry);
                int x1 = (int)floorf(-m[1] + rx) + 1, y1 = (int)floorf(-m[3] + ry) + 1;     // right/top neighbours of the last tap
                ok = (x1 - x0 + 1 <= TILE_PITCH) && (y1 - y0 + 1 <= TILE_ROWS);
                if (ok) { sh.red[0][0] = (float)(x1 - x0 + 1); sh.red[0][1] = (float)(y1 - y0 + 1); }
            } else if (n_bad == 0 && !(m[0] < INFINITY)) {
                ok = 2;                                      // nothing to blur in this block
            }
            sh.x0 = x0; sh.y0 = y0; sh.ok = ok;
        }
        __syncthreads();
        if (sh.ok != 1) return;
        // 2. stage {T[x,y], T[x+1,y]} as float16, addressing per texture.py:274-283 (repeat or clamp)
        const int tw = (int)sh.red[0][0], th = (int)sh.red[0][1];
        const int x0 = sh.x0, y0 = sh.y0;
        const uint8_t* data = (const uint8_t*)bg.data;
        const int comps = bg.components;
        for (int idx = tid; idx < tw*th; idx += nthreads) {
            const int ty = idx / tw, tx = idx - ty*tw;
            const int j = wrap_texel(y0 + ty, bg.height, bg.repeat_y);
            const int i0 = wrap_texel(x0 + tx, bg.width, bg.repeat_x), i1 = wrap_texel(x0 + tx + 1, bg.width, bg.repeat_x);
            const uint8_t* p0 = data + ((long)j*bg.width + i0)*comps;
            const uint8_t* p1 = data + ((long)j*bg.width + i1)*comps;
            half8 v;
            v[0] = (_Float16)(float)p0[0]; v[1] = (_Float16)(float)p0[1]; v[2] = (_Float16)(float)p0[2]; v[3] = (_Float16)0.0f;
            v[4] = (_Float16)(float)p1[0]; v[5] = (_Float16)(float)p1[1]; v[6] = (_Float16)(float)p1[2]; v[7] = (_Float16)0.0f;
            sh.tile[ty*TILE_PITCH + tx] = v;
        }
        __syncthreads();
        (void)any;
    }

    __device__ __forceinline__ static void tap(const half8* tile, float x, float y, float& r, float& g, float& b) {
        const float fx = floorf(x), fy = floorf(y);
        const float ax = x - fx, ay = y - fy;
        const half8* p = tile + (int)fy*TILE_PITCH + (int)fx;
        const half8 lo = p[0], hi = p[TILE_PITCH];
        const float w11 = ax*ay;
        const float w10 = ax - w11, w01 = ay - w11;
        const float w00 = (1.0f - ax) - w01;
        r = fmaf(w00, (float)lo[0], r); g = fmaf(w00, (float)lo[1], g); b = fmaf(w00, (float)lo[2], b);
        r = fmaf(w10, (float)lo[4], r); g = fmaf(w10, (float)lo[5], g); b = fmaf(w10, (float)lo[6], b);
        r = fmaf(w01, (float)hi[0], r); g = fmaf(w01, (float)hi[1], g); b = fmaf(w01, (float)hi[2], b);
        r = fmaf(w11, (float)hi[4], r); g = fmaf(w11, (float)hi[5], g); b = fmaf(w11, (float)hi[6], b);
    }

    __device__ static vec4 blur_tile(const RenderArgs& a, const Tex& bg, const State& s, const Shared& sh) {
        const float xr = s.xc - (float)sh.x0, yr = s.yc - (float)sh.y0;
        // displacement of tap k in texels: d_k * intensity * (scale.x*w, h)   (glsl.hpp gtexture)
        const float ax = s.pre.intensity*((float)bg.height/(float)bg.width)*(float)bg.width;
        const float ay = s.pre.intensity*(float)bg.height;
        float r = 0.0f, g = 0.0f, b = 0.0f;
#pragma unroll
        for (int k = 0; k < 10; k++) tap(sh.tile, fmaf(a.tap_x[k], ax, xr), fmaf(a.tap_y[k], ay, yr), r, g, b);
        r = r*2.0f; g = g*2.0f; b = b*2.0f;          // direction 8 == direction 0
        for (int d = 0; d < 71; d += 1) {            // centre tap + directions 1..7
            const int k = 10 + d;
            tap(sh.tile, fmaf(a.tap_x[k], ax, xr), fmaf(a.tap_y[k], ay, yr), r, g, b);
        }
        const float quality = 10.0f, directions = 8.0f;
        return {(r/255.0f)/(quality*directions), (g/255.0f)/(quality*directions), (b/255.0f)/(quality*directions), 91.0f/(quality*directions)};
    }

    __device__ static vec4 run(const RenderArgs& a, const Frag& f, const State& s, const Shared& sh) {
        if (s.pre.out_of_bounds) {
            const vec3 space = vec3{1.0f, 11.0f, 26.0f}/255.0f;
            return {space.x, space.y, space.z, 0.0f};
        }
        const Tex& bg = f.tex[TEX_BACKGROUND];
        vec4 blurred = (sh.ok == 1) ? blur_tile(a, bg, s, sh) : visualizer_blur_reference(f, s.pre);
        return visualizer_post(f, s.pre, blurred);
    }
};

// A background the tile path can stage: unorm8 RGB/RGBA, bilinear
inline bool visualizer_tile_applicable(const Tex& bg) {
    return bg.data && bg.dtype == DT_U8 && bg.components >= 3 && bg.filter == FILTER_LINEAR;
}

}  // namespace sf
