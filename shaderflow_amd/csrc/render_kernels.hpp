// render_kernels.hpp — gfx950 kernels for the pixel half of the path:
//   K6  k_render<F>            one thread per target pixel, RGBA8/float target   (shader.py:388-405)
//   K8  k_resolve              final.glsl as its own pass                        (fragment/final.glsl:1-33)
//   K6+K8 k_render_resolve<F,S> shade S x S supersamples in the lanes of a quad, quantise each to the
//                              iScreen RGBA8 value, resolve with DPP lane exchanges, write RGB8 only
//   K7  the sampler lives in glsl.hpp (generic) and in BlurTile below (LDS-staged background tile)
//   K9  encoder hand-off       the resolve kernels write rows top-down on request (RenderArgs/ResolveArgs.top_down)
//
// Launch geometry: every block of the fused kernel owns 128 consecutive output pixels of one row, so
// it writes 384 contiguous, 128-byte aligned bytes of the RGB8 frame (whole cache lines from a single
// XCD's L2); blocks are numbered so that the eight XCDs (block b runs on XCD b % 8, observed) own
// interleaved 8-row bands and neighbouring rows of background texels stay in the same 4 MiB L2.
#pragma once

#include "fragments.hpp"

#include <hip/hip_fp16.h>

namespace sf {

struct RenderArgs {
    Uniforms u;
    Tex tex[TEX_SLOTS];
    int wr, hr;                      // shaded resolution (scene.render_resolution, scene.py:372-375)
    int w, h;                        // output resolution (fused kernels)
    int subsample;                   // final.glsl kernel size
    int out_components, out_dtype;   // unfused target format
    void* out;
    long out_frame_stride;           // bytes between consecutive frames of a batch (grid.z)
    // tape (batched export) mode: per-frame uniforms and audio textures come from device memory
    const FrameDyn* dyn;
    const float* tape_spectrogram; long spectrogram_stride;   // floats per frame
    const float* tape_bars;          // sqrt(column/1000) per frame (k_visualizer_bars), same stride; visualizer only
    const float* tape_waveform; long waveform_stride;
    int frame0;
    // radial-blur tap table of visualizer.frag:26-31 (unit displacements cos/sin(angle)*walk)
    float tap_x[81], tap_y[81];
    // visualizer.frag terms that depend on uniforms only: per frame on the device (tape), or by value
    const VisualizerConsts* vis_consts;
    VisualizerConsts vis;
    int has_vis;
    int identity_camera;             // glsl.hpp camera_is_identity(u): iCamera.gluv == gluv exactly
    int axis_camera;                 // glsl.hpp camera_is_axis_aligned(u): iCamera.gluv.x is a function of gluv.x alone, .y of gluv.y (zoom, pan, dolly: no rotation)
    float aspect;                    // iResolution.x/iResolution.y (iAspectRatio, shaderflow.glsl:16), divided once on the host
    float bg_scale_x;                // background.height/background.width (gtexture, shaderflow.glsl:166-167), divided once on the host
    int tile_pitch, tile_rows;       // geometry of the LDS tile when it is a launch parameter (VisualizerShader<0, …>)
    int top_down;                    // K9: write the RGB8 frame rows top-down (the encoder's `vflip`, exporting.py:103, done here)
    float inv_wr, inv_hr;            // RN(1/wr), RN(1/hr) when the host verified glsl.hpp pixel_centre() for them, else 0
    int quads;                       // a bound sampler is mipmapped: k_render covers 32 x 2 pixels per wave as 2 x 2 quads (64 x 4 blocks), helper lanes shaded
    unsigned* tile_misses;           // when set (sfx_ctx_tile_misses): counts the blocks of the LDS-tiled kernels that fell back to the generic taps
    int affine_camera;               // iCamera.gluv is an AFFINE function of gluv (a camera rolled about its forward axis, zoomed, panned): cam_affine holds it
    float cam_affine[6];             // iCamera.gluv = (cam_affine[0], [1]) + gluv.x*([2], [3]) + gluv.y*([4], [5]) — for BOUNDS only (a block's window), never for a sample
#ifdef SF_SECTION_TIMERS
    unsigned long long* timers;      // profiling builds (tools/variants.sh): per-section shader-clock sums, see SF_TICK
#endif
};

// Fingerprint of the kernel-argument layout: the offsets of RenderArgs' members and the sizes of the blocks inside it, plus the
// build switches that change it. The library exports it (sfx_abi_layout) and every code object of a translated fragment carries
// the value it was compiled with (sfx_jit_layout): a same-size change — two members swapped — is refused at load time too.
constexpr unsigned long long layout_mix(unsigned long long h, unsigned long long v) { return (h ^ v)*1099511628211ull; }
constexpr unsigned long long render_args_layout() {
    unsigned long long h = 1469598103934665603ull;
#define SF_LAYOUT_MEMBER(m) h = layout_mix(h, (unsigned long long)__builtin_offsetof(RenderArgs, m))
    SF_LAYOUT_MEMBER(u); SF_LAYOUT_MEMBER(tex); SF_LAYOUT_MEMBER(wr); SF_LAYOUT_MEMBER(hr); SF_LAYOUT_MEMBER(w); SF_LAYOUT_MEMBER(h);
    SF_LAYOUT_MEMBER(subsample); SF_LAYOUT_MEMBER(out_components); SF_LAYOUT_MEMBER(out_dtype); SF_LAYOUT_MEMBER(out);
    SF_LAYOUT_MEMBER(out_frame_stride); SF_LAYOUT_MEMBER(dyn); SF_LAYOUT_MEMBER(tape_spectrogram); SF_LAYOUT_MEMBER(spectrogram_stride);
    SF_LAYOUT_MEMBER(tape_bars); SF_LAYOUT_MEMBER(tape_waveform); SF_LAYOUT_MEMBER(waveform_stride); SF_LAYOUT_MEMBER(frame0);
    SF_LAYOUT_MEMBER(tap_x); SF_LAYOUT_MEMBER(tap_y); SF_LAYOUT_MEMBER(vis_consts); SF_LAYOUT_MEMBER(vis); SF_LAYOUT_MEMBER(has_vis);
    SF_LAYOUT_MEMBER(identity_camera); SF_LAYOUT_MEMBER(axis_camera); SF_LAYOUT_MEMBER(aspect); SF_LAYOUT_MEMBER(bg_scale_x); SF_LAYOUT_MEMBER(tile_pitch);
    SF_LAYOUT_MEMBER(tile_rows); SF_LAYOUT_MEMBER(top_down); SF_LAYOUT_MEMBER(inv_wr); SF_LAYOUT_MEMBER(inv_hr); SF_LAYOUT_MEMBER(quads); SF_LAYOUT_MEMBER(tile_misses);
    SF_LAYOUT_MEMBER(affine_camera); SF_LAYOUT_MEMBER(cam_affine);
#undef SF_LAYOUT_MEMBER
    h = layout_mix(h, sizeof(RenderArgs)); h = layout_mix(h, sizeof(Uniforms)); h = layout_mix(h, sizeof(Tex));
    h = layout_mix(h, sizeof(FrameDyn)); h = layout_mix(h, sizeof(VisualizerConsts));
    h = layout_mix(h, (unsigned long long)__builtin_offsetof(Uniforms, user)); h = layout_mix(h, (unsigned long long)__builtin_offsetof(Uniforms, iAudioVolume));
    h = layout_mix(h, TEX_SLOTS); h = layout_mix(h, USER_SLOTS);
#ifdef SF_SECTION_TIMERS
    h = layout_mix(h, 0x74696d657273ull);
#endif
    return h;
}

// Section timers of profiling builds: every wave adds the shader-clock cycles it spent since its previous mark to
// timers[section]; compiled out otherwise.
#ifdef SF_SECTION_TIMERS
#define SF_TIMER_ROWS 8192
__device__ __forceinline__ unsigned long long sf_clock() {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define SF_TICK_INIT() unsigned long long sf_tick_last = sf_clock()
#define SF_TICK(a, section) do { const unsigned long long sf_now = sf_clock(); \
    if ((threadIdx.x & 63) == 0 && (a).timers) atomicAdd(&(a).timers[(((blockIdx.x + blockIdx.z*gridDim.x)*8 + (threadIdx.x >> 6)) % SF_TIMER_ROWS)*8 + section], sf_now - sf_tick_last); \
    sf_tick_last = sf_clock(); } while (0)
#else
#define SF_TICK_INIT() do {} while (0)
#define SF_TICK(a, section) do {} while (0)
#endif

__device__ __forceinline__ void frame_view(const RenderArgs& a, int frame, Uniforms& u, Tex* tex) {
    u = a.u;
    for (int k = 0; k < TEX_HISTORY; k++) tex[k] = a.tex[k];        // history slots are read in place (Frag::history)
    if (a.dyn) {
        const FrameDyn d = a.dyn[a.frame0 + frame];
        u.iTime = d.iTime; u.iTau = d.iTau; u.iFrame = d.iFrame;
        u.iAudioVolume = d.iAudioVolume; u.iAudioVolumeIntegral = d.iAudioVolumeIntegral; u.iAudioSTD = d.iAudioSTD;
        u.iSpectrogramOffset = d.iSpectrogramOffset;
    }
    if (a.tape_spectrogram) tex[TEX_SPECTROGRAM].data = a.tape_spectrogram + (long)(a.frame0 + frame)*a.spectrogram_stride;
    if (a.tape_waveform) tex[TEX_WAVEFORM].data = a.tape_waveform + (long)(a.frame0 + frame)*a.waveform_stride;
}

__device__ __forceinline__ void store_target(const RenderArgs& a, long frame, int i, int j, vec4 c) {
    char* base = (char*)a.out + frame*a.out_frame_stride;
    const long pix = (long)j*a.wr + i;
    const int n = a.out_components;
    if (a.out_dtype == DT_U8) {
        if (n == 4) { ((uint32_t*)base)[pix] = pack_rgba8(c); return; }
        uint8_t* p = (uint8_t*)base + pix*n;
        p[0] = (uint8_t)unorm8(c.x);
        if (n > 1) p[1] = (uint8_t)unorm8(c.y);
        if (n > 2) p[2] = (uint8_t)unorm8(c.z);
    } else if (a.out_dtype == DT_F16) {                          // round to nearest even, as a GL half-float attachment stores
        _Float16* p = (_Float16*)base + pix*n;
        p[0] = (_Float16)c.x;
        if (n > 1) p[1] = (_Float16)c.y;
        if (n > 2) p[2] = (_Float16)c.z;
        if (n > 3) p[3] = (_Float16)c.w;
    } else {
        float* p = (float*)base + pix*n;
        p[0] = c.x;
        if (n > 1) p[1] = c.y;
        if (n > 2) p[2] = c.z;
        if (n > 3) p[3] = c.w;
    }
}


// ---- shader policies -------------------------------------------------------------------------------------
// A kernel shades its supersamples in two passes around one block-cooperative step: pre() per sample,
// setup() once per block (may stage LDS and synchronise), run() per sample.
template <int FRAGMENT> struct PlainShader {
    static constexpr int BLOCK_W = 64, BLOCK_H = 4;      // unfused block shape
    static constexpr int MIN_WAVES_PER_SIMD = 1;
#ifndef PLAIN_FUSED_ROWS
#define PLAIN_FUSED_ROWS 4
#endif
    static constexpr int FUSED_ROWS = PLAIN_FUSED_ROWS;  // output rows a lane group walks in the fused kernel (S >= 2)
    static constexpr int THREAD_ROWS = 1;                // output rows covered side by side by the block's threads (S >= 2)
    static constexpr int BLOCK_PX = 128;                 // output pixels of one row per block of the fused kernel (S >= 2)
    struct State {};
    struct Shared {};
    __device__ static void pre(const RenderArgs&, const Frag&, bool, State&) {}
    template <int N> __device__ static void setup(const RenderArgs&, const Tex*, const Frag&, State (&)[N], const bool (&)[N], Shared&, int) {}
    __device__ static vec4 run(const RenderArgs&, const Frag& f, const State&, const Shared&) { return shade<FRAGMENT>(f); }
};

// ---- K6: generic unfused render --------------------------------------------------------------------
// (the bodies are functions so that the code object of a run-time translated fragment can wrap them in kernels with
// C names, jit_runtime.hpp)
// Shaders that take screen-space derivatives (translated fragments calling dFdx/dFdy/fwidth) declare `QUADS = true`: the
// lanes of a wave then cover 32 x 2 pixels as sixteen 2 x 2 quads in consecutive lanes (lane bit 0 = x, bit 1 = y inside the
// quad), every lane of a quad is shaded whether or not its pixel exists, and a derivative is a DPP quad_perm difference.
template <class S, class = void> struct shader_uses_quads { static constexpr bool value = false; };
template <class S> struct shader_uses_quads<S, decltype((void)S::QUADS)> { static constexpr bool value = S::QUADS; };

// Shaders may opt out of the verified-reciprocal pixel centres (glsl.hpp pixel_centre) at compile time: `FAST_CENTRES = false`
template <class S, class = void> struct shader_fast_centres { static constexpr bool value = true; };
template <class S> struct shader_fast_centres<S, decltype((void)S::FAST_CENTRES)> { static constexpr bool value = S::FAST_CENTRES; };

// Rows a lane walks where the kernels give it one sample at a time (the 1x fused kernel; the unfused kernel's blocks stacked
// vertically): 1 unless the shader's per-block setup wants more samples to pay for it (`ROWS_1X`, the tiled translated fragments)
template <class S, class = void> struct shader_rows_1x { static constexpr int value = 1; };
template <class S> struct shader_rows_1x<S, decltype((void)S::ROWS_1X)> { static constexpr int value = S::ROWS_1X; };

template <class SHADER>
__device__ __forceinline__ void render_body(const RenderArgs& a) {
    __shared__ typename SHADER::Shared shared;
    constexpr bool QUADS = shader_uses_quads<SHADER>::value;
    constexpr int R = shader_rows_1x<SHADER>::value;     // rows a thread walks: the block covers BLOCK_W x (BLOCK_H*R) pixels
    static_assert(!(QUADS && R > 1), "the quad layout shades one pixel per lane");
    // shaders written for 64 x 4 blocks that shade one pixel per lane take the quad layout at run time too (a.quads: a mipmapped sampler)
    constexpr bool QUADS_POSSIBLE = (SHADER::BLOCK_W == 64 && SHADER::BLOCK_H == 4 && R == 1);
    const bool quads = QUADS || (QUADS_POSSIBLE && a.quads);
    int lx = threadIdx.x, ly = threadIdx.y;
    if (quads) {
        static_assert(!QUADS || (SHADER::BLOCK_W == 64 && SHADER::BLOCK_H == 4), "the quad layout is written for 64 x 4 blocks");
        const int tid = ly*64 + lx, wave = tid >> 6, lane = tid & 63;
        lx = (((lane >> 2) << 1) | (lane & 1)) + 32*(wave & 1);
        ly = ((lane >> 1) & 1) + 2*(wave >> 1);
    }
    const int i = blockIdx.x*SHADER::BLOCK_W + lx;
    const int j = (blockIdx.y*SHADER::BLOCK_H + ly)*R;
    Uniforms u; Tex tex[TEX_HISTORY];
    frame_view(a, blockIdx.z, u, tex);
    Frag f; f.u = &u; f.tex = tex; f.history = a.tex + TEX_HISTORY;
    typename SHADER::State state[R];
    bool valid[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        valid[r] = (i < a.wr) && (j + r < a.hr);
        make_varyings(f, i, j + r, a.wr, a.hr, a.aspect, shader_fast_centres<SHADER>::value ? a.inv_wr : 0.0f, shader_fast_centres<SHADER>::value ? a.inv_hr : 0.0f);
        SHADER::pre(a, f, valid[r], state[r]);
    }
    // the thread that owns the block's last valid pixel (thread 0 owns the first): corner samples for monotone shaders
    const int i_last = min((int)(blockIdx.x + 1)*SHADER::BLOCK_W, a.wr) - 1 - (int)blockIdx.x*SHADER::BLOCK_W;
    const int j_last = (min((int)(blockIdx.y + 1)*SHADER::BLOCK_H*R, a.hr) - 1 - (int)blockIdx.y*SHADER::BLOCK_H*R)/R;
    SHADER::template setup<R>(a, tex, f, state, valid, shared, j_last*SHADER::BLOCK_W + i_last);
#pragma unroll
    for (int r = 0; r < R; r++) {
        if constexpr (R > 1) make_varyings(f, i, j + r, a.wr, a.hr, a.aspect, shader_fast_centres<SHADER>::value ? a.inv_wr : 0.0f, shader_fast_centres<SHADER>::value ? a.inv_hr : 0.0f);
        if (quads || valid[r]) {                                          // quad layout: helper lanes run too — their values feed the neighbours' differences
            const vec4 colour = SHADER::run(a, f, state[r], shared);
            if (valid[r]) store_target(a, blockIdx.z, i, j + r, colour);
        }
    }
}
template <class SHADER>
__global__ __launch_bounds__(SHADER::BLOCK_W*SHADER::BLOCK_H, SHADER::MIN_WAVES_PER_SIMD) void k_render(const RenderArgs a) { render_body<SHADER>(a); }

// ---- K8: final.glsl as a pass ------------------------------------------------------------------------
struct ResolveArgs {
    Tex screen;                      // RGBA8, linear, clamp (scene.py:192-194)
    int w, h, subsample;
    uint8_t* out;                    // RGB8 rows bottom-up
    long screen_frame_stride, out_frame_stride;    // bytes between frames of a batch (grid.z)
    int top_down;                    // rows of `out` top-down instead of GL's bottom-up
};

__device__ __forceinline__ vec3 final_glsl(const Tex& screen, vec2 astuv, vec2 resolution, int kernel) {
    if (kernel == 1) return rgb(texture(screen, astuv));                                     // :6-10
    vec3 accumulator = {0.0f, 0.0f, 0.0f};
    vec2 pixel_size = vec2{1.0f, 1.0f}/resolution;                                           // :17
    vec2 corner = astuv - (pixel_size/2.0f);                                                 // :20
    vec2 origin = corner + (pixel_size/(float)kernel)/2.0f;                                  // :21
    for (int x = 0; x < kernel; x++) {
        for (int y = 0; y < kernel; y++) {
            vec2 offset = (pixel_size/(float)kernel)*vec2{(float)x, (float)y};               // :25
            accumulator = accumulator + rgb(texture(screen, origin + offset));               // :26
        }
    }
    return accumulator/(float)(kernel*kernel);                                               // :31
}

#ifdef SF_UNIT_RESOLVE       // a non-template kernel lives in ONE translation unit of the library (launch_resolve.hip)
__global__ __launch_bounds__(256) void k_resolve(const ResolveArgs a) {
    const int i = blockIdx.x*64 + threadIdx.x;
    const int j = blockIdx.y*4 + threadIdx.y;
    if (i >= a.w || j >= a.h) return;
    vec2 centre = {((float)i + 0.5f)/(float)a.w, ((float)j + 0.5f)/(float)a.h};
    vec2 astuv = gluv2stuv(centre*2.0f - 1.0f);
    Tex screen = a.screen;
    screen.data = (const char*)a.screen.data + (long)blockIdx.z*a.screen_frame_stride;
    vec3 c = final_glsl(screen, astuv, vec2{(float)a.w, (float)a.h}, a.subsample);
    const int row = a.top_down ? (a.h - 1 - j) : j;
    uint8_t* p = a.out + (long)blockIdx.z*a.out_frame_stride + ((long)row*a.w + i)*3;
    p[0] = (uint8_t)unorm8(c.x); p[1] = (uint8_t)unorm8(c.y); p[2] = (uint8_t)unorm8(c.z);
}
#endif

__device__ __forceinline__ float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }   // sf::clamp(x, 0, 1) for every non-NaN x (the sign of a zero is squared away by its users)
__device__ __forceinline__ float smoothstep01(float t) { t = clamp01(t); return t*t*(3.0f - 2.0f*t); }   // sf::smoothstep(0, 1, t)

// ---- DPP helpers ---------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
// quad_perm broadcasts: every lane of a quad reads lane q of its quad
__device__ __forceinline__ uint32_t quad_lane0(uint32_t v) { return dpp_u32<0x00>(v); }
__device__ __forceinline__ uint32_t quad_lane1(uint32_t v) { return dpp_u32<0x55>(v); }
__device__ __forceinline__ uint32_t quad_lane2(uint32_t v) { return dpp_u32<0xAA>(v); }
__device__ __forceinline__ uint32_t quad_lane3(uint32_t v) { return dpp_u32<0xFF>(v); }

__device__ __forceinline__ float unorm_channel(uint32_t q, int shift) { return unorm8_to_float((float)((q >> shift) & 255u)); }

// The iScreen texel a supersample becomes (RGBA8 unorm), then final.glsl (:13-31) over the S x S block of one
// output pixel, for ONE colour channel (bits [shift, shift+8) of the packed texels). q[] holds the block in texel
// order q[y*S + x]. Taps land on texel centres (S == K) or on the centre of 2x2 texels (S == 2K); the reference's
// bilinear weights are then 1 or 1/4 up to the rounding of the tap coordinate.
template <int S, int K>
__device__ __forceinline__ float resolve_channel(const uint32_t (&q)[S*S], int shift) {
    float acc = 0.0f;
    if constexpr (K == S) {                          // taps on texel centres, x-major like final.glsl:23-28
#pragma unroll
        for (int x = 0; x < S; x++)
#pragma unroll
            for (int y = 0; y < S; y++) acc = acc + unorm_channel(q[y*S + x], shift);
        return acc/(float)(S*S);
    } else {
        constexpr int G = S/K;                       // == 2: each tap is the mean of a 2x2 group
        static_assert(G == 2, "fused resolve needs S == K or S == 2K");
#pragma unroll
        for (int x = 0; x < K; x++) {
#pragma unroll
            for (int y = 0; y < K; y++) {
                const float tap = bilerp(0.25f, 0.25f, 0.25f, 0.25f,
                                         unorm_channel(q[(y*G)*S + x*G], shift), unorm_channel(q[(y*G)*S + x*G + 1], shift),
                                         unorm_channel(q[(y*G + 1)*S + x*G], shift), unorm_channel(q[(y*G + 1)*S + x*G + 1], shift));
                acc = (K == 1) ? tap : acc + tap;
            }
        }
        return (K == 1) ? acc : acc/(float)(K*K);
    }
}

template <int S>
__device__ __forceinline__ uint32_t resolve_channel_any(const uint32_t (&q)[S*S], int kernel, int shift) {
    float c;
    if constexpr (S == 1) c = resolve_channel<1, 1>(q, shift);
    else if constexpr (S == 2) c = (kernel == 2) ? resolve_channel<2, 2>(q, shift) : resolve_channel<2, 1>(q, shift);
    else c = (kernel == 4) ? resolve_channel<4, 4>(q, shift) : resolve_channel<4, 2>(q, shift);
    return unorm8(c);
}

__host__ __device__ inline bool fused_supported(int s, int kernel) {
    return (s == 1 && kernel == 1) || (s == 2 && (kernel == 2 || kernel == 1)) || (s == 4 && (kernel == 4 || kernel == 2));
}

// Block id → tile. The dispatcher places block b on XCD b % 8 (observed, never relied on for correctness): giving
// XCD k the k-th eighth of the (row-major) tile list keeps one horizontal band of the frame — and the band of
// background texels under it — in one XCD's 4 MiB L2.
__device__ __forceinline__ int xcd_band_order(int b, int nblocks) {
    return (nblocks % 8 == 0) ? (b % 8)*(nblocks/8) + (b/8) : b;
}

// Writes one row segment of 128 RGB8 pixels (384 B) staged in LDS with 16-byte stores
// 16 bytes of a finished frame from LDS to HBM as a NON-TEMPORAL store (`global_store_dwordx4 ... nt`): frames are written once and
// read by nobody on the device — a batch is 1.5 GB, hundreds of times the L2 — so the lines need not be kept; the store-bound
// kernels gain 10-15 % (k_separable_runs: MusicBars 164 000 -> 189 000 frames/s at 4K 2x)
typedef uint32_t StreamWords __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_store16(void* dst, const void* src) { __builtin_nontemporal_store(*(const StreamWords*)src, (StreamWords*)dst); }
__device__ __forceinline__ void store_rgb_row(uint8_t* out_row, int x0, int w, const uint8_t* staged, int tid, int nthreads, int block_px) {
    const int npix = (w - x0 < block_px) ? (w - x0) : block_px;
    const int nbytes = npix*3;
    uint8_t* dst = out_row + (long)x0*3;
    if ((nbytes & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
        for (int k = tid; k < nbytes/16; k += nthreads) stream_store16((uint4*)dst + k, (const uint4*)staged + k);
    } else {
        for (int k = tid; k < nbytes; k += nthreads) dst[k] = staged[k];
    }
}

// ---- K6+K8 fused ------------------------------------------------------------------------------------------
// S == 1: one lane per output pixel (256 threads = 128 px x 2 rows).
// S == 2: four lanes (a quad) per output pixel, one supersample each.
// S == 4: a quad per output pixel, each lane owns a 2x2 group of supersamples.
// After shading, every lane of a quad receives the quad's packed RGBA8 texels through DPP quad_perm moves and
// lane c of the quad resolves colour channel c (lane 3 idles); lane 0 collects the three bytes.
template <class SHADER, int S>
__device__ __forceinline__ void render_resolve_body(const RenderArgs& a) {
    constexpr int LANES = (S == 1) ? 1 : 4;
    constexpr int GROUP = (S*S)/LANES;               // supersamples of one pixel owned by one lane: 1, 1, 4
    constexpr int G = (S == 4) ? 2 : 1;              // side of that group
    // Output rows per block: S == 1 packs 2 rows of 128 pixels into its 256 threads; otherwise every quad walks
    // SHADER::FUSED_ROWS vertically adjacent pixels, which amortises the shader's per-block setup (LDS staging)
    constexpr int TROWS = (S == 1) ? 2 : SHADER::THREAD_ROWS;             // rows covered by different threads
    constexpr int WALK = (S == 1) ? shader_rows_1x<SHADER>::value : SHADER::FUSED_ROWS;   // pixels a lane group visits one after the other
    constexpr int ROWS = TROWS*WALK;
    constexpr int PER_LANE = GROUP*WALK;
    constexpr int BPX = (S == 1) ? 128 : SHADER::BLOCK_PX;               // output pixels of a row per block (multiple of 16)
    __shared__ __attribute__((aligned(16))) uint8_t staged[ROWS][BPX*3];

    Uniforms u; Tex tex[TEX_HISTORY];
    frame_view(a, blockIdx.z, u, tex);
    const int blocks_x = (a.w + BPX - 1)/BPX;
    const int tile = xcd_band_order(blockIdx.x, gridDim.x);
    const int bx = tile % blocks_x, by = tile / blocks_x;
    const int tid = threadIdx.x;
    const int p = tid / LANES, sub = tid % LANES;
    const int prow = p / BPX;                        // 0 .. TROWS-1
    const int px = bx*BPX + (p % BPX), py0 = by*ROWS + prow*WALK;

    __shared__ typename SHADER::Shared shared;
    uint32_t mine[PER_LANE];
    typename SHADER::State state[PER_LANE];
    bool valid[PER_LANE];
    Frag f; f.u = &u; f.tex = tex; f.history = a.tex + TEX_HISTORY;
    SF_TICK_INIT();
#pragma unroll
    for (int n = 0; n < PER_LANE; n++) {
        const int r = n / GROUP, m = n % GROUP;
        const int gx = (sub & 1)*G + (m % G), gy = (sub >> 1)*G + (m / G);       // position inside the S x S block
        valid[n] = (px < a.w) && (py0 + r < a.h);
        make_varyings(f, px*S + gx, (py0 + r)*S + gy, a.wr, a.hr, a.aspect, shader_fast_centres<SHADER>::value ? a.inv_wr : 0.0f, shader_fast_centres<SHADER>::value ? a.inv_hr : 0.0f);
        SHADER::pre(a, f, valid[n], state[n]);
    }
    SF_TICK(a, 0);                                   // varyings + pre
    // the thread whose LAST sample is the block's top-right valid supersample (thread 0's first one is the bottom-left)
    const int p_last = min(BPX - 1, a.w - 1 - bx*BPX);
    const int prow_last = (min(ROWS, a.h - by*ROWS) - 1)/WALK;
    const int corner_tid = (prow_last*BPX + p_last)*LANES + (LANES - 1);
    SHADER::template setup<PER_LANE>(a, tex, f, state, valid, shared, corner_tid);
    SF_TICK(a, 1);                                   // setup (window reduction + LDS staging)
#pragma unroll
    for (int n = 0; n < PER_LANE; n++) {
        uint32_t q = 0;
        if (valid[n]) {
            if constexpr (PER_LANE > 1) {            // with one sample per lane the varyings of pass 1 are still live
                const int r = n / GROUP, m = n % GROUP;
                const int gx = (sub & 1)*G + (m % G), gy = (sub >> 1)*G + (m / G);
                make_varyings(f, px*S + gx, (py0 + r)*S + gy, a.wr, a.hr, a.aspect, shader_fast_centres<SHADER>::value ? a.inv_wr : 0.0f, shader_fast_centres<SHADER>::value ? a.inv_hr : 0.0f);
            }
            q = pack_rgba8(SHADER::run(a, f, state[n], shared));
        }
        mine[n] = q;
    }
    SF_TICK(a, 2);                                   // run (blur + post)

    if constexpr (S == 1) {
#pragma unroll
        for (int r = 0; r < WALK; r++) {
            const uint32_t block[1] = {mine[r]};
            if (valid[r]) {
                uint8_t* s = &staged[prow*WALK + r][(p % BPX)*3];
                s[0] = (uint8_t)resolve_channel_any<S>(block, a.subsample, 0);
                s[1] = (uint8_t)resolve_channel_any<S>(block, a.subsample, 8);
                s[2] = (uint8_t)resolve_channel_any<S>(block, a.subsample, 16);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < WALK; r++) {
            uint32_t block[S*S];
#pragma unroll
            for (int m = 0; m < GROUP; m++) {
                const uint32_t v = mine[r*GROUP + m];
                const uint32_t l0 = quad_lane0(v), l1 = quad_lane1(v), l2 = quad_lane2(v), l3 = quad_lane3(v);
                const int ox = m % G, oy = m / G;
                block[(0*G + oy)*S + 0*G + ox] = l0;     // lane sub: gx = (sub&1)*G + ox, gy = (sub>>1)*G + oy
                block[(0*G + oy)*S + 1*G + ox] = l1;
                block[(1*G + oy)*S + 0*G + ox] = l2;
                block[(1*G + oy)*S + 1*G + ox] = l3;
            }
            const uint32_t channel = resolve_channel_any<S>(block, a.subsample, 8*(sub < 3 ? sub : 0));
            const uint32_t green = quad_lane1(channel), blue = quad_lane2(channel);
            if (valid[r*GROUP] && sub == 0) {
                uint8_t* s = &staged[prow*WALK + r][(p % BPX)*3];
                s[0] = (uint8_t)channel; s[1] = (uint8_t)green; s[2] = (uint8_t)blue;
            }
        }
    }
    __syncthreads();
    uint8_t* frame = (uint8_t*)a.out + (long)blockIdx.z*a.out_frame_stride;
    // a full-width block of a frame whose rows are whole 16-byte groups: all rows leave in one sweep of 16-byte stores (the block
    // ends one store latency after its resolve; row after row, the other waves hold the block's registers and LDS meanwhile)
    constexpr int GROUPS = BPX*3/16;
    if (bx*BPX + BPX <= a.w && (a.w*3) % 16 == 0 && ((uintptr_t)frame & 15) == 0) {
        const int rows_here = min(ROWS, a.h - by*ROWS);
        for (int e = tid; e < ROWS*GROUPS; e += (int)blockDim.x) {
            const int r = e/GROUPS, c = e - r*GROUPS;
            if (r < rows_here) {
                const int y = by*ROWS + r;
                uint8_t* row = frame + (long)(a.top_down ? a.h - 1 - y : y)*a.w*3 + (long)bx*BPX*3;
                stream_store16((uint4*)row + c, (const uint4*)&staged[0][0] + e);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int y = by*ROWS + r;
            if (y < a.h) store_rgb_row(frame + (long)(a.top_down ? a.h - 1 - y : y)*a.w*3, bx*BPX, a.w, staged[r], tid, blockDim.x, BPX);
        }
    }
    SF_TICK(a, 3);                                   // resolve + store
}
template <class SHADER, int S>
__global__ __launch_bounds__(4*SHADER::BLOCK_PX*SHADER::THREAD_ROWS, SHADER::MIN_WAVES_PER_SIMD) void k_render_resolve(const RenderArgs a) {
    render_resolve_body<SHADER, S>(a);
}

}  // namespace sf
