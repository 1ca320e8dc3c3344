// visualizer_kernels.hpp — visualizer.frag with its radial blur (examples/basic/shaders/visualizer.frag:21-33)
// evaluated from an LDS-staged tile of the background texture.
//
// The blur is 1 + 9x10 bilinear taps of the background per supersample: ~3.0 G taps per 4K 2xSSAA frame, so this
// loop IS the frame time (FP32 VALU bound, not HBM bound). Every tap of every supersample of
// a block falls inside one small window of the background, so the block:
//   1. reduces the bounding box of its centre taps (wave shuffles + LDS),
//   2. stages that window once — texel addressing (repeat / clamp) per texture.py:274-283 — in the bilinear
//      "difference basis" of each texel cell, as float32 (byte values and their differences are exact):
//          A = T00,  B = T10 - T00,  C = T01 - T00,  D = T00 - T10 - T01 + T11        (per channel)
//      so that a tap at fractional position (ax, ay) of the cell is  A + ax*B + ay*C + (ax*ay)*D ;
//      48 bytes per cell: three ds_read_b128 per tap (a float16 cell with v_fma_mix_f32 was measured slower:
//      v_fma_mix issues at half the rate of v_fma_f32 on gfx950, tools/ubench_valu.hip);
//   3. runs the taps in tile-local texel coordinates: positions advance by VGPR adds, 2 v_fract, the LDS byte
//      offset formed in float arithmetic and converted once, 1 weight product and 12 v_fma/v_add = 24 VALU
//      instructions per tap.
// Direction 8 of the float-counter loop coincides with direction 0 to 3e-9 texel (SURVEY.md §7 hard part 4), so
// direction 0 is evaluated once and counted twice: 81 taps instead of 91.
// The four axis-aligned directions are summed one texel cell at a time (runs_axis): the ten equally spaced taps of
// such a direction cross at most ceil(9*step) cell boundaries, and inside a cell their sum has a closed form.
// Everything that depends on uniforms only (sin/cos of iTime, pow of the volume, …) is evaluated once per frame
// (k_visualizer_consts, or by the host for single launches) instead of once per supersample.
//
// Accuracy contract: tap POSITIONS differ from the generic chain (glsl.hpp stexture) by the rounding of an affine
// re-association (≤ 1e-4 texel), the bilinear sum is re-associated and scaled by 1/255 once instead of per texel:
// ≤ 1e-6 relative on the blurred colour, far inside the 1 LSB pixel tolerance. Everything outside the blur is the
// same code as the generic kernel. If the window does not fit the tile the block falls back to the generic taps.
#pragma once

#include "render_kernels.hpp"

namespace sf {


#ifdef SF_UNIT_VISUALIZER_TABLES       // non-template kernels live in ONE translation unit of the library (launch_visualizer_strip.hip)
__global__ void k_visualizer_consts(const FrameDyn* __restrict__ dyn, int frame0, int nframes, VisualizerConsts* __restrict__ out) {
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= nframes) return;
    const FrameDyn d = dyn[frame0 + k];
    out[frame0 + k] = visualizer_consts(d.iTime, d.iAudioVolume, d.iAudioSTD);
}

// Bar heights per spectrogram texel: visualizer.frag:45 takes sqrt(texel/1000) of the column texel a fragment looks up; the
// value depends on the texel only, so the tape path evaluates it once per (frame, bin, channel) — the same IEEE division
// and square root, hoisted out of the 33 M fragments of a frame.
__global__ void k_visualizer_bars(const float* __restrict__ columns, long n, float* __restrict__ bars) {
    const long k = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (k < n) bars[k] = sf::sqrt(columns[k]/1000.0f);
}
#endif

// TILE_PITCH: cells per tile row (48 B each): 128 covers a 128-pixel block without supersampling, 80 is enough when
// the block's 128 pixels are 2x or 4x supersampled (the window is then ~64 cells wide) and lets more blocks share a CU.
// UNFUSED_W x UNFUSED_H: the block of the unfused kernel. 128 x 2 samples need a 128 x 10 tile (61 KB: two blocks per CU, five
// staged cells per sample); 64 x 8 samples need 64 x 15 cells (46 KB: three blocks of 512 threads, two cells per sample).
template <int TILE_PITCH, int TILE_ROWS, int MIN_WAVES, int ROWS_PER_BLOCK = 1, int THREAD_ROWS_PER_BLOCK = 1, int BLOCK_PIXELS = 128,
          int UNFUSED_W = 128, int UNFUSED_H = 2>
struct VisualizerShader {
    static constexpr int BLOCK_PX = BLOCK_PIXELS;        // output pixels of a row per block of the fused kernel (S >= 2)
    static constexpr int FUSED_ROWS = ROWS_PER_BLOCK;    // rows a lane group walks; one LDS window serves FUSED_ROWS*THREAD_ROWS output rows
    static constexpr int THREAD_ROWS = THREAD_ROWS_PER_BLOCK;
    static constexpr int BLOCK_W = UNFUSED_W, BLOCK_H = UNFUSED_H;     // unfused block shape (render_kernels.hpp k_render)
    static constexpr int MIN_WAVES_PER_SIMD = MIN_WAVES;
    // this kernel runs at its issue and dependency limits: the division's latency is hidden and the shorter sequence of
    // glsl.hpp pixel_centre() gains nothing here (A/B on one box: -0.4 %), so it keeps the division
    static constexpr bool FAST_CENTRES = false;

    struct State {
        VisualizerPre pre;
        float xc, yc;                // centre tap in texel space (u*w - 0.5, v*h - 0.5)
    };
    // TILE_PITCH == 0: the tile geometry is a launch parameter (RenderArgs::tile_pitch/tile_rows) and the cells live in
    // dynamic LDS — for output/background combinations whose window does not fit the fixed tile (capi launch_fused)
    static constexpr bool DYNAMIC_TILE = (TILE_PITCH == 0);
    struct Shared {
        float4 cells[DYNAMIC_TILE ? 1 : TILE_ROWS*TILE_PITCH*3];   // 48-byte cells {Ar Ag Ab Br} {Bg Bb Cr Cg} {Cb Dr Dg Db}
        float red[5][16];
        VisualizerConsts consts;
        int x0, y0, tw, th, ok;
    };
    __device__ __forceinline__ static float4* tile_of(Shared& sh) {
        if constexpr (DYNAMIC_TILE) { extern __shared__ __attribute__((aligned(16))) float4 sf_dynamic_tile[]; return sf_dynamic_tile; }
        else return sh.cells;
    }
    __device__ __forceinline__ static const float4* tile_of(const Shared& sh) { return tile_of(const_cast<Shared&>(sh)); }
    __device__ __forceinline__ static int pitch_of(const RenderArgs& a) { if constexpr (DYNAMIC_TILE) return a.tile_pitch; else return TILE_PITCH; }
    __device__ __forceinline__ static int rows_of(const RenderArgs& a) { if constexpr (DYNAMIC_TILE) return a.tile_rows; else return TILE_ROWS; }
    // bytes between tile rows, as the float the offset arithmetic uses (a literal for the fixed tile)
    __device__ __forceinline__ static float row_bytes(const RenderArgs& a) { if constexpr (DYNAMIC_TILE) return (float)(a.tile_pitch*48); else return (float)(TILE_PITCH*48); }

    __device__ static VisualizerConsts frame_consts(const RenderArgs& a, const Frag& f) {
        if (a.vis_consts) return a.vis_consts[a.frame0 + blockIdx.z];
        if (a.has_vis) return a.vis;
        return visualizer_consts(f.u->iTime, f.u->iAudioVolume, f.u->iAudioSTD);
    }

    __device__ static void pre(const RenderArgs& a, const Frag& f, bool, State& s) {
        const VisualizerConsts c = frame_consts(a, f);
        s.pre = visualizer_pre(f, c, a.identity_camera != 0);
        const Tex& bg = f.tex[TEX_BACKGROUND];
        // same chain as stexture() → texture() for the centre tap (glsl.hpp)
        vec2 scale = {a.bg_scale_x, 1.0f};                           // (float)bg.height/(float)bg.width, the same division on the host
        vec2 st = gluv2stuv(stuv2gluv(s.pre.bg)*scale);
        s.xc = st.x*(float)bg.width - 0.5f;
        s.yc = st.y*(float)bg.height - 0.5f;
    }

    // Minimum over the wave with DPP row operations (quad swaps, half-mirror, mirror, then the two row broadcasts):
    // six v_min_f32_dpp and one v_readlane instead of six ds_bpermute round trips.
    template <int CTRL, int ROW_MASK> __device__ __forceinline__ static float min_dpp(float v) {
        const int moved = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
        return fminf(v, __int_as_float(moved));
    }
    __device__ static float wave_min(float v) {
        v = min_dpp<0xB1, 0xF>(v);       // quad_perm [1,0,3,2]
        v = min_dpp<0x4E, 0xF>(v);       // quad_perm [2,3,0,1]
        v = min_dpp<0x141, 0xF>(v);      // row_half_mirror
        v = min_dpp<0x140, 0xF>(v);      // row_mirror: every lane holds the minimum of its row of 16
        v = min_dpp<0x142, 0xA>(v);      // row_bcast15 into rows 1 and 3
        v = min_dpp<0x143, 0xC>(v);      // row_bcast31 into rows 2 and 3: lane 63 holds the wave minimum
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    }

    template <int N>
    __device__ static void setup(const RenderArgs& a, const Tex* tex, const Frag& f, State (&s)[N], const bool (&valid)[N], Shared& sh, int corner_tid) {
        const Tex& bg = tex[TEX_BACKGROUND];
        const int tid = threadIdx.y*blockDim.x + threadIdx.x, nthreads = blockDim.x*blockDim.y;
        const int wave = tid >> 6, nwaves = (nthreads + 63) >> 6;
        const VisualizerConsts c = frame_consts(a, f);
        // tap radius in texels: |cos|,|sin| <= 1, walk <= 1.0000001 (visualizer.frag:26-28), + one step (0.1 of the
        // radius): the run loops leave their position one step past the last tap
        const float rx = fabsf(c.intensity*a.bg_scale_x*(float)bg.width)*1.101f + 0.001f;
        const float ry = fabsf(c.intensity*(float)bg.height)*1.101f + 0.001f;
        SF_TICK_INIT();
        int ok = 0, x0 = 0, y0 = 0, tw = 0, th = 0;                  // block-uniform window [x0, x0+tw) x [y0, y0+th)
        if (a.identity_camera) {
            // 1a. With the identity camera the centre tap is a monotone (non-decreasing, rounding included) function of the
            // pixel column for x and of the pixel row for y, so the block's two corner samples bound all of them: no
            // reduction, one barrier, and every thread derives the window itself.
            if (tid == 0) { sh.red[0][0] = s[0].xc; sh.red[0][1] = s[0].yc; }
            if (tid == corner_tid) { sh.red[1][0] = s[N - 1].xc; sh.red[1][1] = s[N - 1].yc; }
            __syncthreads();
            const float x_lo = fminf(sh.red[0][0], sh.red[1][0]), x_hi = fmaxf(sh.red[0][0], sh.red[1][0]);
            const float y_lo = fminf(sh.red[0][1], sh.red[1][1]), y_hi = fmaxf(sh.red[0][1], sh.red[1][1]);
            if ((c.intensity == c.intensity) && fabsf(x_lo) < 1e8f && fabsf(x_hi) < 1e8f && fabsf(y_lo) < 1e8f && fabsf(y_hi) < 1e8f && rx < 64.0f && ry < 64.0f) {
                x0 = (int)floorf(x_lo - rx); y0 = (int)floorf(y_lo - ry);
                tw = (int)floorf(x_hi + rx) - x0 + 1;
                th = (int)floorf(y_hi + ry) - y0 + 1;
                ok = (tw <= pitch_of(a)) && (th <= rows_of(a));
            }
            if (tid == 0) { sh.x0 = x0; sh.y0 = y0; sh.ok = ok; sh.consts = c; }      // read by run() after the staging barrier
        } else if (a.affine_camera) {
            // 1a'. A camera rolled about its forward axis (zoomed, panned): iCamera.gluv is an affine function of gluv, so the centre
            // taps of a block are bounded by those of its four corner samples — which every thread derives itself from the block's
            // place in the grid: no reduction and no barrier before the staging (round 4 spent 29 % of a rolled launch in the block-wide
            // minimum and its three barriers). The affine map is the HOST's (three get_camera evaluations); the samples themselves keep
            // the generic chain, and the bound carries a margin for the difference (a twentieth of a texel + the rounding of the chain).
            int i_lo, i_hi, j_lo, j_hi;                               // the block's sample rectangle
            if (blockDim.y == 1 && gridDim.y == 1) {                  // the fused kernel (render_resolve_body): a 1-D grid of pixel tiles
                const int ss = a.wr/a.w;
                const int bpx = (ss == 1) ? 128 : BLOCK_PX, rows = (ss == 1) ? 2*shader_rows_1x<VisualizerShader>::value : THREAD_ROWS*FUSED_ROWS;
                const int blocks_x = (a.w + bpx - 1)/bpx;
                const int tile_index = xcd_band_order(blockIdx.x, gridDim.x);
                const int bx = tile_index % blocks_x, by = tile_index / blocks_x;
                i_lo = bx*bpx*ss; i_hi = min((bx + 1)*bpx, a.w)*ss - 1; j_lo = by*rows*ss; j_hi = min((by + 1)*rows, a.h)*ss - 1;
            } else {                                                  // the unfused kernel (render_body)
                const int rows = BLOCK_H*shader_rows_1x<VisualizerShader>::value;
                i_lo = blockIdx.x*BLOCK_W; i_hi = min(i_lo + BLOCK_W, a.wr) - 1; j_lo = blockIdx.y*rows; j_hi = min(j_lo + rows, a.hr) - 1;
            }
            float x_lo = INFINITY, x_hi = -INFINITY, y_lo = INFINITY, y_hi = -INFINITY;
#pragma unroll
            for (int corner = 0; corner < 4; corner++) {
                Frag g = f;
                make_varyings(g, (corner & 1) ? i_hi : i_lo, (corner & 2) ? j_hi : j_lo, a.wr, a.hr, a.aspect);
                const vec2 uv = {fmaf(g.gluv.y, a.cam_affine[4], fmaf(g.gluv.x, a.cam_affine[2], a.cam_affine[0])),
                                 fmaf(g.gluv.y, a.cam_affine[5], fmaf(g.gluv.x, a.cam_affine[3], a.cam_affine[1]))};
                const vec2 bgc = (gluv2stuv(uv) - vec2{0.5f, 0.5f})*c.zoom2 + vec2{0.5f, 0.5f} + vec2{c.off_x, c.off_y};   // visualizer_pre
                const vec2 st = gluv2stuv(stuv2gluv(bgc)*vec2{a.bg_scale_x, 1.0f});                                          // pre(): stexture's chain
                const float xc = st.x*(float)bg.width - 0.5f, yc = st.y*(float)bg.height - 0.5f;
                x_lo = fminf(x_lo, xc); x_hi = fmaxf(x_hi, xc); y_lo = fminf(y_lo, yc); y_hi = fmaxf(y_hi, yc);
            }
            const float slack = 0.05f + 1.0e-5f*(fabsf(x_lo) + fabsf(x_hi) + fabsf(y_lo) + fabsf(y_hi));
            if ((c.intensity == c.intensity) && fabsf(x_lo) < 1e8f && fabsf(x_hi) < 1e8f && fabsf(y_lo) < 1e8f && fabsf(y_hi) < 1e8f && rx < 64.0f && ry < 64.0f) {
                x0 = (int)floorf(x_lo - rx - slack); y0 = (int)floorf(y_lo - ry - slack);
                tw = (int)floorf(x_hi + rx + slack) - x0 + 1;
                th = (int)floorf(y_hi + ry + slack) - y0 + 1;
                ok = (tw <= pitch_of(a)) && (th <= rows_of(a));
            }
            if (tid == 0) { sh.x0 = x0; sh.y0 = y0; sh.tw = tw; sh.th = th; sh.ok = ok; sh.consts = c; }      // read by run() after the staging barrier
        } else {
        // 1b. bounding box of the centre taps (as minima of x, -x, y, -y)
        float lo_x = INFINITY, hi_x = INFINITY, lo_y = INFINITY, hi_y = INFINITY;
        bool bad = !(c.intensity == c.intensity);
#pragma unroll
        for (int n = 0; n < N; n++) {
            if (valid[n] && !s[n].pre.out_of_bounds) {
                bad = bad || !(fabsf(s[n].xc) < 1e8f) || !(fabsf(s[n].yc) < 1e8f);
                lo_x = fminf(lo_x, s[n].xc); hi_x = fminf(hi_x, -s[n].xc);
                lo_y = fminf(lo_y, s[n].yc); hi_y = fminf(hi_y, -s[n].yc);
            }
        }
        lo_x = wave_min(lo_x); hi_x = wave_min(hi_x); lo_y = wave_min(lo_y); hi_y = wave_min(hi_y);
        const int n_bad = __syncthreads_count(bad ? 1 : 0);
        if ((tid & 63) == 0) { sh.red[0][wave] = lo_x; sh.red[1][wave] = hi_x; sh.red[2][wave] = lo_y; sh.red[3][wave] = hi_y; }
        __syncthreads();
        if (tid == 0) {
            float m[4];
            for (int k = 0; k < 4; k++) { m[k] = sh.red[k][0]; for (int w = 1; w < nwaves; w++) m[k] = fminf(m[k], sh.red[k][w]); }
            if (n_bad == 0 && m[0] < INFINITY && rx < 64.0f && ry < 64.0f) {
                x0 = (int)floorf(m[0] - rx); y0 = (int)floorf(m[2] - ry);
                tw = (int)floorf(-m[1] + rx) - x0 + 1;                // cells [x0, x0+tw) hold every tap's floor()
                th = (int)floorf(-m[3] + ry) - y0 + 1;
                ok = (tw <= pitch_of(a)) && (th <= rows_of(a));
            } else if (n_bad == 0 && !(m[0] < INFINITY)) {
                ok = 2;                                               // nothing to blur in this block
            }
            sh.x0 = x0; sh.y0 = y0; sh.tw = tw; sh.th = th; sh.ok = ok;
            sh.consts = c;
        }
        __syncthreads();
        ok = sh.ok; x0 = sh.x0; y0 = sh.y0; tw = sh.tw; th = sh.th;
        }
        SF_TICK(a, 4);                               // window (incl. its barriers)
        if (ok == 0 && tid == 0 && a.tile_misses) atomicAdd(a.tile_misses, 1u);
#ifdef SF_DEBUG_MISS                                                    // tools/variants_par.sh "miss:-DSF_DEBUG_MISS": which blocks leave their tile, and by how much
        if (ok == 0 && tid == 0) printf("block %d: window %d x %d at (%d, %d), tile %d x %d, affine %d\n", (int)blockIdx.x, tw, th, x0, y0, pitch_of(a), rows_of(a), a.affine_camera);
#endif
        if (ok != 1) { __syncthreads(); return; }    // sh.ok is read by run(): publish it like the staged path does
        // 2. stage the cells
        const uint8_t* data = (const uint8_t*)bg.data;
        const int comps = bg.components;
        // one thread per cell of the TILE_PITCH-wide grid (constant divisor); a texel is fetched with ONE unaligned
        // 4-byte load (the allocation is padded, capi sfx_texture_create) and unpacked with v_cvt_f32_ubyteN
        typedef uint32_t unaligned_u32 __attribute__((aligned(1)));
        const int pitch = pitch_of(a);
        float4* const tile = tile_of(sh);
        // fixed tile: one thread per cell of the pitch-wide grid (constant divisor); dynamic tile: waves take rows, lanes columns
        const int idx_end = DYNAMIC_TILE ? th*((tw + 63) & ~63) : pitch*th, row_span = DYNAMIC_TILE ? ((tw + 63) & ~63) : pitch;
        for (int idx = tid; idx < idx_end; idx += nthreads) {
            int ty, tx;
            if constexpr (DYNAMIC_TILE) { ty = idx / row_span; tx = idx - ty*row_span; }      // row_span is a multiple of 64: a wave stays in one row
            else { ty = idx / TILE_PITCH; tx = idx - ty*TILE_PITCH; }
            if (tx >= tw) continue;
            const int j0 = wrap_texel(y0 + ty, bg.height, bg.repeat_y), j1 = wrap_texel(y0 + ty + 1, bg.height, bg.repeat_y);
            const int i0 = wrap_texel(x0 + tx, bg.width, bg.repeat_x), i1 = wrap_texel(x0 + tx + 1, bg.width, bg.repeat_x);
            const uint32_t row0 = (uint32_t)j0*(uint32_t)bg.width, row1 = (uint32_t)j1*(uint32_t)bg.width;
            const uint32_t w00 = *(const unaligned_u32*)(data + (size_t)(row0 + i0)*comps);
            const uint32_t w10 = *(const unaligned_u32*)(data + (size_t)(row0 + i1)*comps);
            const uint32_t w01 = *(const unaligned_u32*)(data + (size_t)(row1 + i0)*comps);
            const uint32_t w11 = *(const unaligned_u32*)(data + (size_t)(row1 + i1)*comps);
            float4 q0, q1, q2;                                        // {Ar Ag Ab Br} {Bg Bb Cr Cg} {Cb Dr Dg Db}
            {
                // byte values and their differences are exact in binary32
                const float r00 = (float)(w00 & 255u), g00 = (float)((w00 >> 8) & 255u), b00 = (float)((w00 >> 16) & 255u);
                const float r10 = (float)(w10 & 255u), g10 = (float)((w10 >> 8) & 255u), b10 = (float)((w10 >> 16) & 255u);
                const float r01 = (float)(w01 & 255u), g01 = (float)((w01 >> 8) & 255u), b01 = (float)((w01 >> 16) & 255u);
                const float r11 = (float)(w11 & 255u), g11 = (float)((w11 >> 8) & 255u), b11 = (float)((w11 >> 16) & 255u);
                q0 = make_float4(r00, g00, b00, r10 - r00);
                q1 = make_float4(g10 - g00, b10 - b00, r01 - r00, g01 - g00);
                q2 = make_float4(b01 - b00, (r00 - r10) - (r01 - r11), (g00 - g10) - (g01 - g11), (b00 - b10) - (b01 - b11));
            }
            float4* cell = tile + (ty*pitch + tx)*3;
            cell[0] = q0; cell[1] = q1; cell[2] = q2;
        }
        SF_TICK(a, 5);                               // staging (loads + conversion + LDS writes)
        __syncthreads();
        SF_TICK(a, 6);                               // barrier after staging
    }

    // One bilinear tap from the staged cells: value = A + ax*B + ay*C + (ax*ay)*D per channel.
    // Issue cost on gfx950 (tools/ubench_valu.hip): v_fma/v_add/v_mul_f32 with VGPR operands 2 cycles per wave64,
    // everything else (v_fract, v_cvt, shifts, v_fma_mix, any VALU with an SGPR operand) 4 cycles — so the float32
    // tile (12 plain fma/add = 24 cycles) beats the float16 tile (12 v_fma_mix = 48 cycles) although it reads 48
    // instead of 32 bytes of LDS per tap, and every per-tap operand is kept in VGPRs.
    // the lerp of one tap whose cell offset (bytes, exact in float) and fractions are known
    __device__ __forceinline__ static void tap_at(const float4* tile, float cell, float ax, float ay, float& r, float& g, float& b) {
        const float4* p = (const float4*)((const char*)tile + (unsigned)cell);
        const float4 q0 = p[0], q1 = p[1], q2 = p[2];
        const float axy = ax*ay;
        r = r + q0.x;           g = g + q0.y;           b = b + q0.z;
        r = fmaf(ax, q0.w, r);  g = fmaf(ax, q1.x, g);  b = fmaf(ax, q1.y, b);
        r = fmaf(ay, q1.z, r);  g = fmaf(ay, q1.w, g);  b = fmaf(ay, q2.x, b);
        r = fmaf(axy, q2.y, r); g = fmaf(axy, q2.z, g); b = fmaf(axy, q2.w, b);
    }

    __device__ __forceinline__ static void tap(const float4* tile, float rowb, float x, float y, float& r, float& g, float& b) {
        const float ax = __builtin_amdgcn_fractf(x), ay = __builtin_amdgcn_fractf(y);
        // byte offset of the cell, in float arithmetic (exact: < 2^24) so that only ONE conversion is needed:
        // (floor(y)*PITCH + floor(x))*48; x, y >= 0 inside the staged window. (Carrying the offset along with the
        // position — one more add, two fewer ops — was measured: 32 fewer VALU per wave, same time.)
        const float cell = fmaf(y - ay, rowb, (x - ax)*48.0f);
        const float4* p = (const float4*)((const char*)tile + (unsigned)cell);
        const float4 q0 = p[0], q1 = p[1], q2 = p[2];
        const float axy = ax*ay;
        r = r + q0.x;           g = g + q0.y;           b = b + q0.z;
        r = fmaf(ax, q0.w, r);  g = fmaf(ax, q1.x, g);  b = fmaf(ax, q1.y, b);
        r = fmaf(ay, q1.z, r);  g = fmaf(ay, q1.w, g);  b = fmaf(ay, q2.x, b);
        r = fmaf(axy, q2.y, r); g = fmaf(axy, q2.z, g); b = fmaf(axy, q2.w, b);
    }

    // The ten taps of an axis-aligned direction, one texel cell at a time. Inside a cell the bilinear basis is fixed and
    // the taps m_j = m + j*s are equally spaced, so a run of n taps sums to
    //     n*A + S*B' + (n*f)*C' + (S*f)*D,   S = sum_j fract(m_j) = n*(a + s*(n-1)/2),
    // (B', C' = the moving / fixed axis terms, f = the fixed fraction): ONE cell fetch and 12 fma per run instead of
    // per tap. n = min(taps left, floor(t) + 1) with t = (1 - a)/s for s > 0 and a/|s| for s < 0 — the number of
    // steps that stay inside the cell. A tap within rounding distance of a cell boundary may be counted on either
    // side: the bilinear surface is continuous there, so the sum moves by ~1e-7. At most 1 + ceil(9|s|) runs are
    // needed (the loop bound is wave-uniform); the tail loop only runs if rounding split a run.
    // One run of one axis direction: state (m = position of the next tap, left = taps left).
    template <bool ALONG_X>
    __device__ __forceinline__ static void one_run(const float4* tile, float rowb, float& m, float& left, float s, float k0, float k1, float hs,
                                                   float f, float fixed_offset, float weight, float& r, float& g, float& b) {
        const float a = __builtin_amdgcn_fractf(m);
        const float cell = fmaf(m - a, ALONG_X ? 48.0f : rowb, fixed_offset);
        const float4* p = (const float4*)((const char*)tile + (unsigned)cell);
        const float4 q0 = p[0], q1 = p[1], q2 = p[2];
        const float n = fminf(floorf(fmaf(a, k1, k0)) + 1.0f, left);
        left = left - n;
        m = fmaf(n, s, m);
        const float nw = n*weight;
        const float sum = nw*(a + fmaf(n, hs, -hs));
        const float wx = ALONG_X ? sum : nw*f, wy = ALONG_X ? nw*f : sum, wxy = sum*f;
        r = fmaf(nw, q0.x, r);   g = fmaf(nw, q0.y, g);   b = fmaf(nw, q0.z, b);
        r = fmaf(wx, q0.w, r);   g = fmaf(wx, q1.x, g);   b = fmaf(wx, q1.y, b);
        r = fmaf(wy, q1.z, r);   g = fmaf(wy, q1.w, g);   b = fmaf(wy, q2.x, b);
        r = fmaf(wxy, q2.y, r);  g = fmaf(wxy, q2.z, g);  b = fmaf(wxy, q2.w, b);
    }

    // The four axis-aligned directions together: every loop iteration advances each of them by one run, which gives the
    // scheduler four independent dependency chains and one loop test. `step` > 0 is the common |step| (texels per tap),
    // `first` the offset of the first tap; direction 0 (+x) carries weight 2 (direction 8 == direction 0).
    __device__ __forceinline__ static void runs_axes(const float4* tile, float rowb, float xr, float yr, float first, float step, float fx, float fy,
                                                     float row_offset, float column_offset, float& r, float& g, float& b) {
        const float inv = fminf(__builtin_amdgcn_rcpf(step), 1.0e6f);
        const float hs = 0.5f*step;
        const int bound = __builtin_amdgcn_readfirstlane(1 + (int)ceilf(9.0f*step + 1.0e-3f));
        float m0 = xr + first, m1 = yr + first, m2 = xr - first, m3 = yr - first;
        float l0 = 10.0f, l1 = 10.0f, l2 = 10.0f, l3 = 10.0f;
        for (int it = 0; it < bound || __any((l0 + l1) + (l2 + l3) > 0.0f); it++) {
            one_run<true>(tile, rowb, m0, l0, step, inv, -inv, hs, fy, row_offset, 2.0f, r, g, b);
            one_run<false>(tile, rowb, m1, l1, step, inv, -inv, hs, fx, column_offset, 1.0f, r, g, b);
            one_run<true>(tile, rowb, m2, l2, -step, 0.0f, inv, -hs, fy, row_offset, 1.0f, r, g, b);
            one_run<false>(tile, rowb, m3, l3, -step, 0.0f, inv, -hs, fx, column_offset, 1.0f, r, g, b);
        }
    }

    __device__ static vec4 blur_tile(const RenderArgs& a, const Tex& bg, const State& s, const Shared& sh) {
        const float xr = s.xc - (float)sh.x0, yr = s.yc - (float)sh.y0;
        // displacement of tap k in texels: d_k * intensity * (scale.x*w, h)   (glsl.hpp gtexture)
        const float ax = sh.consts.intensity*a.bg_scale_x*(float)bg.width;
        float r = 0.0f, g = 0.0f, b = 0.0f;
        // Taps of one direction are (up to 1e-7 relative) an arithmetic progression: walk = 0.1, 0.2, … (:27); stepping
        // with VGPR adds keeps the position update on the 2-cycle path (an fma with the SGPR table entry costs 4).
        // Directions 0/4 and 2/6 run along a texel row / column: their cross-axis displacement is |cos(k*TAU/4)| <= 2e-7
        // of the radius (< 1e-6 texel), which is dropped so that the fixed coordinate is decomposed once per direction.
        const float fy = __builtin_amdgcn_fractf(yr), fx = __builtin_amdgcn_fractf(xr);
        const float4* const tile = tile_of(sh);
        const float rowb = row_bytes(a);
        const float row_offset = (yr - fy)*rowb, column_offset = (xr - fx)*48.0f;
        // axis-aligned directions: closed-form runs, the four directions interleaved (the two axes have the same texel
        // scale up to 1 ulp, see below)
        runs_axes(tile, rowb, xr, yr, a.tap_x[0]*ax, (a.tap_x[1] - a.tap_x[0])*ax, fx, fy, row_offset, column_offset, r, g, b);
        // The four diagonal directions walk (+-k*s, +-k*s) from the centre (|cos| and |sin| of 45 degrees agree to 1 ulp,
        // and so do the texel scales of the two axes: differences below 1e-6 texel are dropped): at every walk step
        // the four taps share two x and two y coordinates, so two fractions and two cell offsets per axis serve all four.
        {
            const float ROW = rowb;
            const float step = (a.tap_x[11] - a.tap_x[10])*ax, first = a.tap_x[10]*ax;      // direction 1 = 45 degrees: both positive
            float xp = xr + first, xm = xr - first, yp = yr + first, ym = yr - first;
#ifndef VIS_DIAG_UNROLL
#define VIS_DIAG_UNROLL 1
#endif
#pragma unroll VIS_DIAG_UNROLL
            for (int w = 0; w < 10; w++) {
                const float axp = __builtin_amdgcn_fractf(xp), axm = __builtin_amdgcn_fractf(xm);
                const float ayp = __builtin_amdgcn_fractf(yp), aym = __builtin_amdgcn_fractf(ym);
                const float cxp = (xp - axp)*48.0f, cxm = (xm - axm)*48.0f;
                const float ryp = (yp - ayp)*ROW, rym = (ym - aym)*ROW;
                tap_at(tile, cxp + ryp, axp, ayp, r, g, b);       // 45
                tap_at(tile, cxm + ryp, axm, ayp, r, g, b);       // 135
                tap_at(tile, cxm + rym, axm, aym, r, g, b);       // 225
                tap_at(tile, cxp + rym, axp, aym, r, g, b);       // 315
                xp = xp + step; xm = xm - step; yp = yp + step; ym = ym - step;
            }
        }
        tap(tile, rowb, xr, yr, r, g, b);                            // centre tap (:19)
        // (sum/255)/(quality*directions) (:32) as one multiplication: part of this path's re-association (≤ 1 ulp)
        const float norm = 1.0f/(255.0f*10.0f*8.0f);
        return {r*norm, g*norm, b*norm, 91.0f/80.0f};
    }

    __device__ static vec4 run(const RenderArgs& a, const Frag& f, const State& s, const Shared& sh) {
        if (s.pre.out_of_bounds) {
            const vec3 space = vec3{1.0f, 11.0f, 26.0f}/255.0f;
            return {space.x, space.y, space.z, 0.0f};
        }
        const Tex& bg = f.tex[TEX_BACKGROUND];
        const VisualizerConsts c = sh.consts;
#if defined(VIS_ABLATE_BLUR)
        vec4 blurred = {s.xc*1e-3f, s.yc*1e-3f, 0.5f, 1.0f};             // ablation builds only (tools/variants.sh)
#else
        SF_TICK_INIT();
        vec4 blurred = (sh.ok == 1) ? blur_tile(a, bg, s, sh) : visualizer_blur_reference(f, s.pre, c);
        SF_TICK(a, 7);                               // blur only
#endif
#if defined(VIS_ABLATE_POST)
        return blurred;
#else
#ifndef VIS_EXACT_POST
        return visualizer_post<true>(f, s.pre, c, blurred, a.tape_bars ? a.tape_bars + (long)(a.frame0 + blockIdx.z)*a.spectrogram_stride : nullptr);
#else
        return visualizer_post<false>(f, s.pre, c, blurred);
#endif
#endif
    }
};

// A background the tile path can stage: unorm8 RGB/RGBA, bilinear
inline bool visualizer_tile_applicable(const Tex& bg) {
    return bg.data && bg.dtype == DT_U8 && bg.components >= 3 && bg.filter == FILTER_LINEAR;
}

}  // namespace sf
