// capi.hip — implementation of include/shaderflow_hip.h (the C-ABI over the gfx950 kernels): contexts, textures, programs (the fragment
// registry, uniform / sampler tables), the render dispatch into the launch units (launch.hpp), the clock sequence and the render
// from a tape. Host-side state only. The read-out rings and peer copies live in capi_readout.hip, the audio plans and the tape in
// capi_audio.hip; host_state.hpp holds what the three share.

#include "launch.hpp"
#include "launch_geometry.hpp"
#include "visualizer_kernels.hpp"
#include "uniform_table.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <algorithm>
#include <vector>

using namespace sf;

// ---------------------------------------------------------------------------------------------------------
// Errors and handles

thread_local std::string g_error;
thread_local std::string g_last_kernel;   // which render kernel instance the last launch on this thread picked (sfx_last_kernel)

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_error = buf;
    return code;
}

extern "C" const char* sfx_last_error(void) { return g_error.c_str(); }
extern "C" const char* sfx_version(void) { return "shaderflow_hip 0.2 (gfx950)"; }
// (defined after the kernel headers) fingerprint of the kernel-argument layout this library was built with
extern "C" uint64_t sfx_abi_layout(void);

thread_local Context* g_launch_ctx = nullptr;               // the context whose program is being launched (scratch owner)

static size_t dtype_size(int dtype) { return dtype == SFX_U8 ? 1 : (dtype == SFX_F32 ? 4 : 2); }

static Tex tex_view(const Texture* t) {
    Tex v{};
    if (t) { v.data = t->data; v.width = t->width; v.height = t->height; v.components = t->components;
             v.dtype = t->dtype; v.filter = t->filter; v.repeat_x = t->repeat_x; v.repeat_y = t->repeat_y; v.mips = t->mips; v.levels = t->levels;
             // the context's filter model: LINEAR unorm8 textures through the fixed-point filter (glsl.hpp texture_fixed8). Every kernel
             // with sampler arithmetic of its own asks for FILTER_LINEAR and so steps aside for the generic ones (mipmapped minification
             // keeps the specification's arithmetic: its llvmpipe form is the oracle's checker switch only)
             if (t->ctx && t->ctx->filter_model == SFX_FILTER_FIXED8 && t->dtype == SFX_U8 && t->filter == SFX_LINEAR) v.filter = FILTER_LINEAR_FIXED8; }
    return v;
}


// ---------------------------------------------------------------------------------------------------------
// Context

// visualizer.frag:23-31 evaluated once in binary32: angles 0, τ/8, … (9 of them, the last one coincides with
// the first), walks 0.1 … 1.0000001. Table order: directions 0..7 (10 taps each), then the centre tap.
static void build_tap_table(float* tx, float* ty) {
    const float quality = 10.0f, directions = 8.0f;
    int dir = 0, n = 0;
    float cx[16][10], cy[16][10];
    int ndir = 0;
    for (float angle = 0.0f; angle < sf::TAU; angle += sf::TAU/directions) {
        int w = 0;
        for (float walk = 1.0f/quality; walk <= 1.001f; walk += 1.0f/quality) {
            if (ndir < 16 && w < 10) { cx[ndir][w] = sf::cos(angle)*walk; cy[ndir][w] = sf::sin(angle)*walk; }
            w++;
        }
        ndir++;
    }
    (void)dir;
    for (int d = 0; d < 8; d++) for (int w = 0; w < 10; w++) { tx[n] = cx[d][w]; ty[n] = cy[d][w]; n++; }
    tx[n] = 0.0f; ty[n] = 0.0f;                                   // [80]: the centre tap
}

// Blocks of the LDS-tiled visualizer kernels whose tap window did not fit their tile (they ran the generic taps instead) since the
// previous call; counting starts with the first call. A tuning aid: results are the same either way.
extern "C" int sfx_ctx_tile_misses(sfx_handle h, unsigned long long* blocks) {
    CTX_OR_FAIL(c, h);
    USE_DEVICE(c);
    unsigned count = 0;
    if (!c->tile_misses) {
        HIP_TRY(hipMalloc((void**)&c->tile_misses, sizeof(unsigned)));
    } else {
        HIP_TRY(hipMemcpyAsync(&count, c->tile_misses, sizeof count, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    HIP_TRY(hipMemsetAsync(c->tile_misses, 0, sizeof(unsigned), c->stream));       // ordered with the launches that count into it
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (blocks) *blocks = count;
    return SFX_OK;
}

extern "C" int sfx_ctx_create(int device_id, void* stream, sfx_handle* out) {
    if (!out) return fail(SFX_E_INVALID, "null output");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(SFX_E_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
    if (device_id < 0 || device_id >= count) return fail(SFX_E_INVALID, "device %d out of range (%d devices)", device_id, count);
    Context* c = new Context();
    c->magic = MAGIC_CTX;
    c->device = device_id;
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(hipGetDeviceProperties(&c->prop, device_id));
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else { HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    for (auto& e : c->events) HIP_TRY(hipEventCreate(&e));
    build_tap_table(c->tap_x, c->tap_y);
    // (the copy streams are chosen on first use: peer windows, or a read-out that HSA refuses. Chosen HERE, before the tape's audio
    // stream exists, they cost the light kernels the overlap of audio and render — MusicBars 188 000 → 129 000 frames/s; the
    // read-out rings, which needed them early, no longer run on streams at all)
    *out = handle_of(c);
    return SFX_OK;
}

extern "C" int sfx_ctx_info(sfx_handle h, sfx_ctx_info_t* info) {
    CTX_OR_FAIL(c, h);
    if (!info) return fail(SFX_E_INVALID, "null info");
    memset(info, 0, sizeof *info);
    snprintf(info->device_name, sizeof info->device_name, "%s", c->prop.name);
    snprintf(info->gcn_arch, sizeof info->gcn_arch, "%s", c->prop.gcnArchName);
    info->device_id = c->device;
    info->compute_units = c->prop.multiProcessorCount;
    info->max_texture_dim = 65536;
    info->wavefront_size = c->prop.warpSize;
    info->total_memory = (int64_t)c->prop.totalGlobalMem;
    info->lds_per_cu = (int64_t)c->prop.maxSharedMemoryPerMultiProcessor;
    return SFX_OK;
}

extern "C" int sfx_ctx_synchronize(sfx_handle h) {
    CTX_OR_FAIL(c, h);
    USE_DEVICE(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SFX_OK;
}

extern "C" int sfx_ctx_output_top_down(sfx_handle h, int enabled) {
    CTX_OR_FAIL(c, h);
    c->top_down = enabled ? 1 : 0;
    return SFX_OK;
}

extern "C" int sfx_ctx_filter_model(sfx_handle h, int model) {
    CTX_OR_FAIL(c, h);
    if (model != SFX_FILTER_SPEC && model != SFX_FILTER_FIXED8) return fail(SFX_E_INVALID, "filter model %d", model);
    c->filter_model = model;
    return SFX_OK;
}

extern "C" int sfx_ctx_destroy(sfx_handle h) {
    CTX_OR_FAIL(c, h);
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (auto& e : c->events) hipEventDestroy(e);
    hipFree(c->vis_tables); hipFree(c->vis_bars); hipFree(c->resolve_tables); hipFree(c->tile_misses); hipFree(c->multipass_taps);
    readout_release(c);                                             // peer copier, copy streams, engine records (capi_readout.hip)
    if (c->own_stream) hipStreamDestroy(c->stream);
    c->magic = 0;
    delete c;
    return SFX_OK;
}

extern "C" int sfx_event_record(sfx_handle h, int slot) {
    CTX_OR_FAIL(c, h);
    if (slot < 0 || slot >= 64) return fail(SFX_E_INVALID, "event slot %d", slot);
    USE_DEVICE(c);
    HIP_TRY(hipEventRecord(c->events[slot], c->stream));
    return SFX_OK;
}

extern "C" int sfx_event_elapsed_ms(sfx_handle h, int a, int b, float* ms) {
    CTX_OR_FAIL(c, h);
    if (a < 0 || a >= 64 || b < 0 || b >= 64 || !ms) return fail(SFX_E_INVALID, "event slots");
    USE_DEVICE(c);
    HIP_TRY(hipEventSynchronize(c->events[b]));
    HIP_TRY(hipEventElapsedTime(ms, c->events[a], c->events[b]));
    return SFX_OK;
}

extern "C" int sfx_device_alloc(sfx_handle h, size_t nbytes, void** ptr) {
    CTX_OR_FAIL(c, h);
    USE_DEVICE(c);
    HIP_TRY(hipMalloc(ptr, nbytes ? nbytes : 16));
    return SFX_OK;
}
extern "C" int sfx_device_free(sfx_handle h, void* ptr) {
    CTX_OR_FAIL(c, h);
    USE_DEVICE(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto& readout : c->readouts) readout.second(readout.first);  // a frame buffer may still be being read out (engine copies: no HIP stream)
    HIP_TRY(hipFree(ptr));
    return SFX_OK;
}


extern "C" int sfx_device_read(sfx_handle h, const void* dptr, void* host, size_t nbytes) {
    CTX_OR_FAIL(c, h);
    USE_DEVICE(c);
    HIP_TRY(hipMemcpyAsync(host, dptr, nbytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SFX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Textures

static void forget_texture(struct Texture* t);                      // defined after Program

extern "C" int sfx_texture_create(sfx_handle h, int width, int height, int components, int dtype, sfx_handle* out) {
    CTX_OR_FAIL(c, h);
    if (!out || width < 1 || height < 1 || components < 1 || components > 4) return fail(SFX_E_INVALID, "texture %dx%dx%d", width, height, components);
    if (dtype != SFX_U8 && dtype != SFX_F32 && dtype != SFX_U16 && dtype != SFX_F16) return fail(SFX_E_UNSUPPORTED, "texture dtype %d", dtype);
    if (width > 65536 || height > 65536) return fail(SFX_E_TOO_LARGE, "texture size too large for this context: (%d, %d) > 65536", width, height);
    USE_DEVICE(c);
    Texture* t = new Texture();
    t->magic = MAGIC_TEX; t->ctx = c;
    t->width = width; t->height = height; t->components = components; t->dtype = dtype;
    t->nbytes = (size_t)width*height*components*dtype_size(dtype);
    hipError_t e = hipMalloc(&t->data, t->nbytes + 16);
    if (e != hipSuccess) { delete t; return fail(SFX_E_HIP, "hipMalloc(%zu): %s", t->nbytes, hipGetErrorString(e)); }
    HIP_TRY(hipMemsetAsync(t->data, 0, t->nbytes + 16, c->stream));
    *out = handle_of(t);
    return SFX_OK;
}

extern "C" int sfx_texture_params(sfx_handle h, int filter, int repeat_x, int repeat_y) {
    Texture* t = get<Texture>(h, MAGIC_TEX);
    if (!t) return fail(SFX_E_INVALID, "invalid texture handle");
    if (filter < SFX_NEAREST || filter > SFX_NEAREST_MIPMAP_NEAREST) return fail(SFX_E_INVALID, "texture filter %d", filter);
    t->filter = filter; t->repeat_x = !!repeat_x; t->repeat_y = !!repeat_y;
    return SFX_OK;
}

int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SFX_OK : fail(SFX_E_HIP, "kernel launch: %s", hipGetErrorString(e));
}

// One level of the chain from the one above it — what glGenerateMipmap does on the implementation the goldens come from (Mesa renders
// every level as a LINEAR-filtered, edge-clamped blit of the previous one: tests/golden/mip.npz holds its levels): texel (i, j) of the
// w x h level samples the W x H level at ((i + ½)·W/w, (j + ½)·H/h) — the 2 x 2 box mean where an extent halves exactly, two-texel
// taps that skip texels where an odd extent is floored. unorm formats round the float result to nearest.
__global__ void k_mip_level(Tex src, void* dst, int w, int h) {
    const int i = blockIdx.x*64 + threadIdx.x, j = blockIdx.y*4 + threadIdx.y;
    if (i >= w || j >= h) return;
    // the tap's texel coordinates, exact (an exact halving gives weights of exactly ½: the box mean); edge-clamped
    const double ub = ((double)i + 0.5)*(double)src.width/(double)w - 0.5, vb = ((double)j + 0.5)*(double)src.height/(double)h - 0.5;
    const double fu = ::floor(ub), fv = ::floor(vb);
    const float ax = (float)(ub - fu), ay = (float)(vb - fv);
    const int i0 = wrap_texel((int)fu, src.width, 0), i1 = wrap_texel((int)fu + 1, src.width, 0);
    const int j0 = wrap_texel((int)fv, src.height, 0), j1 = wrap_texel((int)fv + 1, src.height, 0);
    const vec4 t00 = texel(src, i0, j0), t10 = texel(src, i1, j0), t01 = texel(src, i0, j1), t11 = texel(src, i1, j1);
    const float nx = 1.0f - ax, ny = 1.0f - ay, w00 = nx*ny, w10 = ax*ny, w01 = nx*ay, w11 = ax*ay;
    const float v[4] = {bilerp(w00, w10, w01, w11, t00.x, t10.x, t01.x, t11.x), bilerp(w00, w10, w01, w11, t00.y, t10.y, t01.y, t11.y),
                        bilerp(w00, w10, w01, w11, t00.z, t10.z, t01.z, t11.z), bilerp(w00, w10, w01, w11, t00.w, t10.w, t01.w, t11.w)};
    const size_t at = ((size_t)j*w + i)*src.components;
    for (int k = 0; k < src.components; k++) {
        if (src.dtype == DT_U8) ((uint8_t*)dst)[at + k] = (uint8_t)unorm8(v[k]);
        else if (src.dtype == DT_F32) ((float*)dst)[at + k] = v[k];
        else if (src.dtype == DT_F16) ((_Float16*)dst)[at + k] = (_Float16)v[k];
        else { float q = v[k] > 0.0f ? v[k] : 0.0f; q = q < 1.0f ? q : 1.0f; ((uint16_t*)dst)[at + k] = (uint16_t)::rintf(q*65535.0f); }
    }
}

// box.texture.build_mipmaps() of texture.py:277-278: (re)builds levels 1… from the CURRENT level 0. As in OpenGL, and as the
// reference uses it, a later sfx_texture_write changes level 0 only — the chain is as old as the last call of this function.
extern "C" int sfx_texture_build_mipmaps(sfx_handle h) {
    Texture* t = get<Texture>(h, MAGIC_TEX);
    if (!t) return fail(SFX_E_INVALID, "invalid texture handle");
    USE_DEVICE(t->ctx);
    int levels = 1;
    for (int w = t->width, hh = t->height; w > 1 || hh > 1; levels++) { w = w > 1 ? w >> 1 : 1; hh = hh > 1 ? hh >> 1 : 1; }
    if (levels == 1) return SFX_OK;                                   // a 1 x 1 texture is its own chain
    if (!t->mips) {
        Tex whole = tex_view(t); whole.levels = levels;
        size_t bytes = 0;
        for (int w = t->width, hh = t->height, l = 1; l < levels; l++) { w = w > 1 ? w >> 1 : 1; hh = hh > 1 ? hh >> 1 : 1; bytes += (((size_t)w*hh*texel_bytes(whole)) + 15) & ~(size_t)15; }
        HIP_TRY(hipMalloc(&t->mips, bytes + 16));
    }
    t->levels = levels;
    const Tex whole = tex_view(t);
    for (int l = 1; l < levels; l++) {
        const Tex above = mip_level(whole, l - 1), here = mip_level(whole, l);
        hipLaunchKernelGGL(k_mip_level, dim3((here.width + 63)/64, (here.height + 3)/4), dim3(64, 4), 0, t->ctx->stream, above, (void*)here.data, here.width, here.height);
    }
    return launch_status();
}

// level `level` of the chain (0: the texture) into `data`: max(1, width >> level) x max(1, height >> level) texels, rows bottom-up
extern "C" int sfx_texture_read_level(sfx_handle h, int level, void* data, size_t nbytes) {
    Texture* t = get<Texture>(h, MAGIC_TEX);
    if (!t || !data || level < 0 || level >= t->levels) return fail(SFX_E_INVALID, "invalid texture handle, data or level %d of %d", level, t ? t->levels : 0);
    const Tex v = mip_level(tex_view(t), level);
    if (nbytes != (size_t)v.width*v.height*texel_bytes(v)) return fail(SFX_E_INVALID, "level %d holds %zu bytes, asked for %zu", level, (size_t)v.width*v.height*texel_bytes(v), nbytes);
    USE_DEVICE(t->ctx);
    HIP_TRY(hipMemcpyAsync(data, v.data, nbytes, hipMemcpyDeviceToHost, t->ctx->stream));
    HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    return SFX_OK;
}

extern "C" int sfx_texture_write(sfx_handle h, const void* data, size_t nbytes, int x, int y, int w, int hh) {
    Texture* t = get<Texture>(h, MAGIC_TEX);
    if (!t || !data) return fail(SFX_E_INVALID, "invalid texture handle or data");
    USE_DEVICE(t->ctx);
    const size_t texel = (size_t)t->components*dtype_size(t->dtype);
    if (w == 0 && hh == 0) { x = 0; y = 0; w = t->width; hh = t->height; }
    if (x < 0 || y < 0 || w < 1 || hh < 1 || x + w > t->width || y + hh > t->height) return fail(SFX_E_INVALID, "viewport (%d,%d,%d,%d) outside %dx%d", x, y, w, hh, t->width, t->height);
    if (nbytes != (size_t)w*hh*texel) return fail(SFX_E_INVALID, "texture write of %zu bytes, viewport needs %zu", nbytes, (size_t)w*hh*texel);
    char* dst = (char*)t->data + ((size_t)y*t->width + x)*texel;
    HIP_TRY(hipMemcpy2DAsync(dst, (size_t)t->width*texel, data, (size_t)w*texel, (size_t)w*texel, hh, hipMemcpyHostToDevice, t->ctx->stream));
    HIP_TRY(hipStreamSynchronize(t->ctx->stream));       // the host pointer is only borrowed for the call
    return SFX_OK;
}

extern "C" int sfx_texture_read(sfx_handle h, void* data, size_t nbytes) {
    Texture* t = get<Texture>(h, MAGIC_TEX);
    if (!t || !data) return fail(SFX_E_INVALID, "invalid texture handle or data");
    if (nbytes != t->nbytes) return fail(SFX_E_INVALID, "texture read of %zu bytes, texture holds %zu", nbytes, t->nbytes);
    USE_DEVICE(t->ctx);
    HIP_TRY(hipMemcpyAsync(data, t->data, nbytes, hipMemcpyDeviceToHost, t->ctx->stream));
    HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    return SFX_OK;
}

extern "C" int sfx_texture_device_ptr(sfx_handle h, void** ptr, size_t* nbytes) {
    Texture* t = get<Texture>(h, MAGIC_TEX);
    if (!t) return fail(SFX_E_INVALID, "invalid texture handle");
    if (ptr) *ptr = t->data;
    if (nbytes) *nbytes = t->nbytes;
    return SFX_OK;
}

extern "C" int sfx_texture_destroy(sfx_handle h) {
    Texture* t = get<Texture>(h, MAGIC_TEX);
    if (!t) return fail(SFX_E_INVALID, "invalid texture handle");
    hipSetDevice(t->ctx->device);
    hipStreamSynchronize(t->ctx->stream);
    forget_texture(t);                                              // no program keeps a pointer to it
    hipFree(t->data); hipFree(t->mips);
    t->magic = 0;
    delete t;
    return SFX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Programs

struct RegistryEntry { uint64_t hash; const char* name; };
static const RegistryEntry g_registry[] = {
#include "registry_hashes.inc"
};
static const char* const g_fragment_names[] = {"default", "missing", "visualizer", "bars", "waveform", "multi_child",
                                               "multi_main", "shadertoy", "dynamics", "audio", "multipass", "motionblur",
                                               "life_simulation", "life_visuals", "video", "raymarch", "mandelbrot", "tetration"};
static_assert(sizeof(g_fragment_names)/sizeof(g_fragment_names[0]) == FRAG_COUNT, "one name per fragment");
enum { FRAG_FINAL = 100, FRAG_JIT = 101 };

static int fragment_by_name(const char* name) {
    for (int k = 0; k < FRAG_COUNT; k++) if (!strcmp(name, g_fragment_names[k])) return k;
    if (!strcmp(name, "final")) return FRAG_FINAL;
    return -1;
}

static uint64_t normalised_hash(const char* src) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (const char* p = src; *p;) {
        if (p[0] == '/' && p[1] == '*') { p += 2; while (*p && !(p[0] == '*' && p[1] == '/')) p++; if (*p) p += 2; continue; }
        if (p[0] == '/' && p[1] == '/') { while (*p && *p != '\n') p++; continue; }
        if (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\f' || *p == '\v') { p++; continue; }
        h ^= (uint8_t)*p; h *= 0x100000001b3ull; p++;
    }
    return h;
}

struct JitBinding { std::string name; bool sampler; int slot, count; bool integer; };
struct Program : Object {
    Context* ctx;
    int fragment;
    Uniforms u;
    Texture* samplers[TEX_SLOTS];
    // FRAG_JIT: a code object built by the host from a translated fragment (sfx_program_load)
    hipModule_t module = nullptr;
    hipFunction_t fn_render = nullptr, fn_fused[3] = {nullptr, nullptr, nullptr};     // ssaa 1, 2, 4
    hipFunction_t fn_render_quads = nullptr;       // tiled code objects: the untiled twin of fn_render, whose lanes can form 2 x 2 quads (a mipmapped sampler)
    std::vector<JitBinding> bindings;
    unsigned flags = 0;                                             // sfx_jit_flags of the code object: 1 = takes screen-space derivatives
};

static void forget_texture(Texture* t) {
    for (Program* p : t->ctx->programs)
        for (auto& s : p->samplers) if (s == t) s = nullptr;
}

// scene-defined float uniforms read by a restated fragment: (fragment, name) → user[] slot
struct UserUniform { int fragment; const char* name; int slot; int count; };
static const UserUniform g_user_uniforms[] = {
    {FRAG_DYNAMICS, "iShaderDynamics", 0, 1},
    {FRAG_MOTIONBLUR, "iScreenTemporal", USER_SCREEN_TEMPORAL, 1},      // texture.py:377-379 (<name>Temporal)
    {FRAG_LIFE_SIMULATION, "iLifeSize", USER_LIFE_SIZE, 2},             // texture.py:376 (<name>Size)
    {FRAG_LIFE_SIMULATION, "iLifePeriod", USER_LIFE_PERIOD, 1},         // demo.py:244-246
};
// the texture whose temporal history `<prefix>{t}x0` a fragment samples → slots TEX_HISTORY + t
struct HistoryPrefix { int fragment; const char* prefix; };
static const HistoryPrefix g_history_prefixes[] = {
    {FRAG_MULTIPASS, "iScreen"}, {FRAG_MOTIONBLUR, "iScreen"}, {FRAG_LIFE_SIMULATION, "iLife"}, {FRAG_LIFE_VISUALS, "iLife"}, {FRAG_VIDEO, "iVideo"},
};

struct SamplerName { const char* name; int slot; };
static const SamplerName g_sampler_names[] = { {"background", TEX_BACKGROUND}, {"iSpectrogram", TEX_SPECTROGRAM},
                                               {"iWaveform", TEX_WAVEFORM}, {"child", TEX_CHILD} };


extern "C" int sfx_program_lookup(sfx_handle h, const char* source, sfx_handle* out, int* fallback) {
    CTX_OR_FAIL(c, h);
    if (!source || !out) return fail(SFX_E_INVALID, "null source or output");
    int fragment = fragment_by_name(source);
    if (fragment < 0) {
        // registry stub files shipped with the host package: `#pragma shaderflow_amd kernel(<name>)`
        static const char tag[] = "#pragma shaderflow_amd kernel(";
        if (const char* at = strstr(source, tag)) {
            at += sizeof(tag) - 1;
            const char* end = strchr(at, ')');
            if (end && end - at < 64) fragment = fragment_by_name(std::string(at, end).c_str());
        }
    }
    if (fragment < 0) {
        const uint64_t hash = normalised_hash(source);
        for (const auto& e : g_registry) if (e.hash == hash) { fragment = fragment_by_name(e.name); break; }
    }
    if (fallback) *fallback = (fragment < 0);
    if (fragment < 0) fragment = FRAG_MISSING;                      // shader.py:336-340
    Program* p = new Program();
    p->magic = MAGIC_PROG; p->ctx = c; p->fragment = fragment;
    default_uniforms(p->u);
    for (auto& s : p->samplers) s = nullptr;
    c->programs.push_back(p);
    *out = handle_of(p);
    return SFX_OK;
}

// A fragment translated and compiled by the host (shaderflow_amd/glsl2hip.py → hipcc --genco) — what `opengl.program(vs, fs)`
// (shader.py:324) is for fragments outside the registry. The code object exports the kernels of csrc/jit_runtime.hpp
// (SF_JIT_ENTRY_POINTS) and the size of the RenderArgs it was compiled against.
extern "C" uint64_t sfx_abi_layout(void) { return render_args_layout(); }

extern "C" int sfx_program_load(sfx_handle h, const void* code_object, size_t nbytes, const sfx_binding* bindings, int nbindings, sfx_handle* out) {
    CTX_OR_FAIL(c, h);
    if (!code_object || !nbytes || !out || (nbindings > 0 && !bindings)) return fail(SFX_E_INVALID, "null code object, bindings or output");
    USE_DEVICE(c);
    Program* p = new Program();
    p->magic = MAGIC_PROG; p->ctx = c; p->fragment = FRAG_JIT;
    default_uniforms(p->u);
    for (auto& s : p->samplers) s = nullptr;
    // (the runtime's sticky last error is cleared: launch_status() must not report this failure for a later launch)
    auto bail = [&](int code, const char* what, hipError_t e) { if (p->module) hipModuleUnload(p->module); delete p; (void)hipGetLastError(); return fail(code, "%s: %s", what, hipGetErrorString(e)); };
    hipError_t e = hipModuleLoadData(&p->module, code_object);
    if (e != hipSuccess) { p->module = nullptr; return bail(SFX_E_HIP, "hipModuleLoadData", e); }
    static const char* const fused_names[3] = {"sfx_jit_fused_1", "sfx_jit_fused_2", "sfx_jit_fused_4"};
    if ((e = hipModuleGetFunction(&p->fn_render, p->module, "sfx_jit_render")) != hipSuccess) return bail(SFX_E_INVALID, "code object has no sfx_jit_render", e);
    if (hipModuleGetFunction(&p->fn_render_quads, p->module, "sfx_jit_render_quads") != hipSuccess) { p->fn_render_quads = nullptr; (void)hipGetLastError(); }   // optional
    for (int k = 0; k < 3; k++)
        if ((e = hipModuleGetFunction(&p->fn_fused[k], p->module, fused_names[k])) != hipSuccess) return bail(SFX_E_INVALID, "code object lacks a fused entry point", e);
    hipDeviceptr_t layout = nullptr; size_t layout_bytes = 0; unsigned long long compiled_layout = 0;
    if ((e = hipModuleGetGlobal(&layout, &layout_bytes, p->module, "sfx_jit_layout")) != hipSuccess) return bail(SFX_E_INVALID, "code object has no sfx_jit_layout", e);
    if (layout_bytes != sizeof compiled_layout) { hipModuleUnload(p->module); delete p; return fail(SFX_E_INVALID, "code object predates the layout fingerprint (sfx_jit_layout is %zu bytes): recompile the fragment", layout_bytes); }
    if ((e = hipMemcpy(&compiled_layout, layout, sizeof compiled_layout, hipMemcpyDeviceToHost)) != hipSuccess) return bail(SFX_E_HIP, "reading sfx_jit_layout", e);
    if (compiled_layout != render_args_layout()) {
        hipModuleUnload(p->module); delete p;
        return fail(SFX_E_INVALID, "code object was compiled against another version or build of the kernel headers (argument layout %016llx, library %016llx)", compiled_layout, render_args_layout());
    }
    if (hipModuleGetGlobal(&layout, &layout_bytes, p->module, "sfx_jit_flags") == hipSuccess) {
        if ((e = hipMemcpy(&p->flags, layout, sizeof p->flags, hipMemcpyDeviceToHost)) != hipSuccess) return bail(SFX_E_HIP, "reading sfx_jit_flags", e);
    } else (void)hipGetLastError();
    for (int k = 0; k < nbindings; k++) {
        const sfx_binding& b = bindings[k];
        const int limit = b.sampler ? TEX_SLOTS : USER_SLOTS;
        if (!b.name || b.slot < 0 || b.count < 1 || b.slot + (b.sampler ? 1 : b.count) > limit) { hipModuleUnload(p->module); delete p; return fail(SFX_E_INVALID, "binding %d is out of range", k); }
        p->bindings.push_back({b.name, b.sampler != 0, b.slot, b.count, b.integer != 0});
    }
    c->programs.push_back(p);
    *out = handle_of(p);
    return SFX_OK;
}

// A fragment that takes derivatives needs its 2x2 neighbours in the lanes of a DPP quad. The fused kernel has that layout for
// ssaa == 2 only (the four supersamples of a pixel are the four lanes of a quad, x in bit 0, y in bit 1).
// A mipmapped sampler takes them implicitly (glsl.hpp texture_mipmapped): same rule.
static bool samples_mipmaps(const Program* p) {
    for (int k = 0; k < TEX_SLOTS; k++) if (p->samplers[k] && p->samplers[k]->filter >= SFX_LINEAR_MIPMAP_LINEAR) return true;
    return false;
}
// Under the fixed-point filter model final.glsl's taps go through that filter too (a tap between four texels is NOT their exact mean
// there), so the resolve stays a pass of its own.
static bool fusable(const Program* p, int ssaa) {
    if (p->ctx->filter_model != SFX_FILTER_SPEC) return false;
    return !((p->flags & 1u) || samples_mipmaps(p)) || ssaa == 2;
}
// rows a lane walks in the code object's sfx_jit_render / sfx_jit_fused_1 (shader_rows_1x of its shader policy; older flags words say 0)
static int jit_rows_1x(const Program* p) { const int rows = (int)((p->flags >> 8) & 255u); return rows > 0 ? rows : 1; }
extern "C" int sfx_program_fusable(sfx_handle h, int ssaa) {
    Program* p = get<Program>(h, MAGIC_PROG);
    return (p && p->fragment != FRAG_FINAL && fusable(p, ssaa)) ? 1 : 0;
}

extern "C" const char* sfx_program_name(sfx_handle h) {
    Program* p = get<Program>(h, MAGIC_PROG);
    if (!p) return "";
    if (p->fragment == FRAG_JIT) return "translated";
    return p->fragment == FRAG_FINAL ? "final" : g_fragment_names[p->fragment];
}

extern "C" int sfx_uniform_set(sfx_handle h, const char* name, int type, const void* value, int* known) {
    Program* p = get<Program>(h, MAGIC_PROG);
    if (!p || !name || !value) return fail(SFX_E_INVALID, "invalid program handle, name or value");
    const int counts[] = {1, 1, 1, 2, 3, 4, 4, 9, 16};
    if (type < 0 || type > SFX_T_MAT4) return fail(SFX_E_INVALID, "uniform type %d", type);
    const bool src_int = (type == SFX_T_INT || type == SFX_T_BOOL);
    if (known) *known = 0;
    auto store = [&](char* dst, int count, bool dst_int) {
        const int n = counts[type] < count ? counts[type] : count;
        for (int k = 0; k < n; k++) {
            if (dst_int) ((int*)dst)[k] = src_int ? ((const int*)value)[k] : (int)((const float*)value)[k];
            else ((float*)dst)[k] = src_int ? (float)((const int*)value)[k] : ((const float*)value)[k];
        }
        if (known) *known = 1;
    };
    for (const auto& f : g_uniform_fields)
        if (!strcmp(f.name, name)) { store((char*)&p->u + f.offset, f.count, f.integer); return SFX_OK; }
    for (const auto& uu : g_user_uniforms)
        if (uu.fragment == p->fragment && !strcmp(uu.name, name)) { store((char*)&p->u.user[uu.slot], uu.count, false); return SFX_OK; }
    for (const auto& b : p->bindings)
        if (!b.sampler && b.name == name) { store((char*)&p->u.user[b.slot], b.count, b.integer); return SFX_OK; }
    return SFX_OK;                                                  // inactive uniform: ignored like program.get(name, None)
}

// "background0x0" (texture.py:346-347) and the #define'd plain name (texture.py:355-356) both resolve to the named
// slot; "<prefix>{t}x0" of the fragment's history texture resolves to TEX_HISTORY + t
static int sampler_slot(int fragment, const char* name) {
    std::string base(name);
    int temporal = 0, layer = 0;
    size_t x = base.rfind('x');
    if (x != std::string::npos && x > 0 && x + 1 < base.size()) {
        size_t d = x;
        while (d > 0 && isdigit((unsigned char)base[d - 1])) d--;
        bool tail_digits = true;
        for (size_t k = x + 1; k < base.size(); k++) tail_digits = tail_digits && isdigit((unsigned char)base[k]);
        if (d < x && tail_digits && x - d < 6 && base.size() - x < 7) {
            temporal = atoi(base.substr(d, x - d).c_str());
            layer = atoi(base.substr(x + 1).c_str());
            base = base.substr(0, d);
        }
    }
    for (const auto& hp : g_history_prefixes)
        if (hp.fragment == fragment && base == hp.prefix) return (layer == 0 && temporal < TEX_HISTORY_DEPTH) ? TEX_HISTORY + temporal : -1;
    if (temporal != 0) return -1;                                   // named slots hold the most recent frame only
    for (const auto& s : g_sampler_names) if (base == s.name) return s.slot;
    return -1;
}

extern "C" int sfx_sampler_bind(sfx_handle h, const char* name, sfx_handle tex, int* known) {
    Program* p = get<Program>(h, MAGIC_PROG);
    Texture* t = get<Texture>(tex, MAGIC_TEX);
    if (!p || !name) return fail(SFX_E_INVALID, "invalid program handle or name");
    if (tex && !t) return fail(SFX_E_INVALID, "invalid texture handle");
    int slot = -1;
    if (p->fragment == FRAG_JIT) { for (const auto& b : p->bindings) if (b.sampler && b.name == name) slot = b.slot; }
    else slot = sampler_slot(p->fragment, name);
    if (known) *known = (slot >= 0);
    if (slot >= 0) p->samplers[slot] = t;
    return SFX_OK;
}

// `count` sampler uniforms in one call (a temporal x layers texture matrix after a roll: texture.py:351-381 yields one sampler per
// box and frame); names as for sfx_sampler_bind, unknown ones ignored. The frame loop's per-frame cost is calls, not work.
extern "C" int sfx_sampler_bind_many(sfx_handle h, const char* const* names, const sfx_handle* textures, int count) {
    if (count < 0 || (count > 0 && (!names || !textures))) return fail(SFX_E_INVALID, "sampler table of %d entries", count);
    for (int k = 0; k < count; k++) { int rc = sfx_sampler_bind(h, names[k], textures[k], nullptr); if (rc) return rc; }
    return SFX_OK;
}

// The uniforms of scene.py:687-703 that a frame changes when nothing but the clock moves — iTime, iTau, iDeltatime, iFrame — in one call
extern "C" int sfx_uniform_set_clock(sfx_handle h, float time, float tau, float deltatime, int frame) {
    Program* p = get<Program>(h, MAGIC_PROG);
    if (!p) return fail(SFX_E_INVALID, "invalid program handle");
    p->u.iTime = time; p->u.iTau = tau; p->u.iDeltatime = deltatime; p->u.iFrame = frame;
    return SFX_OK;
}

extern "C" int sfx_program_destroy(sfx_handle h) {
    Program* p = get<Program>(h, MAGIC_PROG);
    if (!p) return fail(SFX_E_INVALID, "invalid program handle");
    auto& live = p->ctx->programs;
    live.erase(std::remove(live.begin(), live.end(), p), live.end());
    if (p->module) { hipSetDevice(p->ctx->device); hipStreamSynchronize(p->ctx->stream); hipModuleUnload(p->module); }
    p->magic = 0;
    delete p;
    return SFX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Launches

static void fill_args(const Program* p, RenderArgs& a) {
    memset(&a, 0, sizeof a);
    a.u = p->u;
    for (int k = 0; k < TEX_SLOTS; k++) a.tex[k] = tex_view(p->samplers[k]);
    memcpy(a.tap_x, p->ctx->tap_x, sizeof a.tap_x);
    memcpy(a.tap_y, p->ctx->tap_y, sizeof a.tap_y);
    // uniform-only terms of visualizer.frag, evaluated once here with the same binary32 operations (sfmath.hpp is
    // host/device); the tape path replaces them per frame on the device (k_visualizer_consts)
    a.vis = visualizer_consts(p->u.iTime, p->u.iAudioVolume, p->u.iAudioSTD);
    a.has_vis = 1;
    a.vis_consts = nullptr;
    a.aspect = p->u.iResolution[0]/p->u.iResolution[1];
    a.identity_camera = camera_is_identity(p->u) ? 1 : 0;
    a.axis_camera = camera_is_axis_aligned(p->u) ? 1 : 0;
    a.bg_scale_x = a.tex[TEX_BACKGROUND].data ? (float)a.tex[TEX_BACKGROUND].height/(float)a.tex[TEX_BACKGROUND].width : 1.0f;
    a.top_down = p->ctx->top_down;
    a.tile_misses = p->ctx->tile_misses;
    a.quads = samples_mipmaps(p) ? 1 : 0;                           // the unfused kernel lays its lanes out as 2 x 2 quads: implicit derivatives
    // a camera rolled about its forward axis (and zoomed / panned): iCamera.gluv is affine in gluv — three evaluations of get_camera give
    // the map the LDS-tiled visualizer kernels bound their blocks' windows with (visualizer_kernels.hpp setup 1a')
    a.affine_camera = 0;
    const Uniforms& u = a.u;
    if (!a.identity_camera && !a.axis_camera && u.iCameraProjection == 0) {
        // (whether the map IS affine is measured, not read off the basis vectors: a rotation by quaternions leaves 1e-17 in components
        // that are zero on paper. Three evaluations define the map, two more — at far corners of the screen — must land on it.)
        auto at = [&](float gx, float gy) { Frag f{}; f.u = &u; f.aspect = a.aspect; f.gluv = vec2{gx, gy}; f.agluv = f.gluv/vec2{a.aspect, 1.0f}; return get_camera(f).gluv; };
        const vec2 origin = at(0.0f, 0.0f), along_x = at(1.0f, 0.0f), along_y = at(0.0f, 1.0f);
        auto predicted = [&](float gx, float gy) { return vec2{origin.x + gx*(along_x.x - origin.x) + gy*(along_y.x - origin.x), origin.y + gx*(along_x.y - origin.y) + gy*(along_y.y - origin.y)}; };
        auto lands = [&](float gx, float gy) { const vec2 is = at(gx, gy), want = predicted(gx, gy); return fabsf(is.x - want.x) < 1e-4f*(1.0f + fabsf(want.x)) && fabsf(is.y - want.y) < 1e-4f*(1.0f + fabsf(want.y)); };
        if (lands(-a.aspect, 1.0f) && lands(a.aspect, -0.8f) && fabsf(origin.x) < 1e6f && fabsf(origin.y) < 1e6f) {
            a.affine_camera = 1;
            a.cam_affine[0] = origin.x; a.cam_affine[1] = origin.y;
            a.cam_affine[2] = along_x.x - origin.x; a.cam_affine[3] = along_x.y - origin.y;
            a.cam_affine[4] = along_y.x - origin.x; a.cam_affine[5] = along_y.y - origin.y;
        }
    }
}

// RN(1/n) if glsl.hpp pixel_centre(i, n, RN(1/n)) equals the IEEE quotient (i + 0.5)/n for every pixel index of an n-pixel axis,
// else 0 (the kernels then divide). Checked exhaustively, once per extent.
static float pixel_centre_reciprocal(int n) {
    static std::mutex lock;
    static std::vector<std::pair<int, float>> known;
    if (n < 1) return 0.0f;
    std::lock_guard<std::mutex> guard(lock);
    for (const auto& k : known) if (k.first == n) return k.second;
    const float inv = 1.0f/(float)n;
    bool same = true;
    for (int i = 0; i < n && same; i++) {
        const float exact = ((float)i + 0.5f)/(float)n, fast = pixel_centre(i, n, inv);
        same = (f2u(exact) == f2u(fast));
    }
    known.push_back({n, same ? inv : 0.0f});
    return known.back().second;
}
#ifdef SF_NO_FAST_CENTRE                                         // A/B builds (tools/variants.sh)
static void set_pixel_centres(RenderArgs& a) { a.inv_wr = 0.0f; a.inv_hr = 0.0f; }
#else
static void set_pixel_centres(RenderArgs& a) { a.inv_wr = pixel_centre_reciprocal(a.wr); a.inv_hr = pixel_centre_reciprocal(a.hr); }
#endif

static bool needs(const RenderArgs& a, int slot) { return a.tex[slot].data != nullptr || (slot == TEX_SPECTROGRAM && a.tape_spectrogram) || (slot == TEX_WAVEFORM && a.tape_waveform); }

static int check_samplers(int fragment, const RenderArgs& a) {
    auto want = [&](int slot, const char* what) { return needs(a, slot) ? SFX_OK : fail(SFX_E_INVALID, "fragment '%s' samples '%s' but no texture is bound", g_fragment_names[fragment], what); };
    int rc = SFX_OK;
    if (fragment == FRAG_VISUALIZER) { if ((rc = want(TEX_BACKGROUND, "background"))) return rc; if ((rc = want(TEX_SPECTROGRAM, "iSpectrogram"))) return rc; return want(TEX_WAVEFORM, "iWaveform"); }
    if (fragment == FRAG_BARS) return want(TEX_SPECTROGRAM, "iSpectrogram");
    if (fragment == FRAG_WAVEFORM) return want(TEX_WAVEFORM, "iWaveform");
    if (fragment == FRAG_MULTI_MAIN) return want(TEX_CHILD, "child");
    if (fragment == FRAG_DYNAMICS) return want(TEX_BACKGROUND, "background");
    if (fragment == FRAG_MULTIPASS) { if (a.u.iLayer == 0) return want(TEX_BACKGROUND, "background"); return want(TEX_HISTORY, "iScreen0x0"); }
    if (fragment == FRAG_MOTIONBLUR) {
        if (a.u.iLayer == 0) return want(TEX_BACKGROUND, "background");
        const int temporal = (int)a.u.user[USER_SCREEN_TEMPORAL];
        if (temporal < 1 || temporal > TEX_HISTORY_DEPTH) return fail(SFX_E_UNSUPPORTED, "motionblur: iScreenTemporal = %d, supported 1..%d", temporal, TEX_HISTORY_DEPTH);
        for (int t = 0; t < temporal; t++) if ((rc = want(TEX_HISTORY + t, "iScreen{t}x0"))) return rc;
        return rc;
    }
    if (fragment == FRAG_LIFE_SIMULATION) {
        if ((int)a.u.user[USER_LIFE_PERIOD] < 1) return fail(SFX_E_INVALID, "life_simulation: iLifePeriod must be >= 1");
        return want(TEX_HISTORY + 1, "iLife1x0");
    }
    if (fragment == FRAG_LIFE_VISUALS) { for (int t = 0; t < 5; t++) if ((rc = want(TEX_HISTORY + t, "iLife{t}x0"))) return rc; return rc; }
    if (fragment == FRAG_VIDEO) return want(TEX_HISTORY, "iVideo");
    return rc;
}

extern "C" const char* sfx_last_kernel(void) { return g_last_kernel.c_str(); }

static int launch_render(int fragment, const RenderArgs& a, int frames, hipStream_t s) {
    using namespace sfl;
    switch (fragment) {
        case FRAG_VISUALIZER: {
            int tw = 0, th = 0;
            {
                const int fast = launch_visualizer_fast(g_launch_ctx, a, 1, frames, s, true);      // identity camera, RGBA8 target, window inside the strip kernel's tile
                if (fast != 0) return fast < 0 ? fast : SFX_OK;
            }
            if (visualizer_tile_applicable(a.tex[TEX_BACKGROUND])) {
                // first choice: 64 x 8 samples per block over a 64 x 15 tile — three 512-thread blocks per CU and two staged cells per
                // sample, against two 256-thread blocks and five cells for the 128 x 2 shape (1080p without SSAA: 8.4 -> see DESIGN §7)
                visualizer_window_bound(a, 64, 8, tw, th);
                if (tw > 0 && tw <= 64 && th <= 15) return render_visualizer_tiled(TILED_R_64x15_WALK8, a, frames, s, 0);
                // sparser outputs (720p over a 1080-row background: 1.3 texels per sample): the same block shape over a tile sized per
                // launch, as long as two blocks share a CU — the 128 x 2 shape would need a 174 x 11 window there (one 256-thread block per CU)
                if (tw > 0 && (size_t)tw*th*48 <= 76*1024) {
                    RenderArgs d = a;
                    d.tile_pitch = tw; d.tile_rows = th;
                    return render_visualizer_tiled(TILED_R_DYNAMIC_WALK8, d, frames, s, (size_t)tw*th*48);
                }
                visualizer_window_bound(a, 128, 2, tw, th);
            }
            if (tw > 0 && tw <= 128 && th <= 10) return render_visualizer_tiled(TILED_R_128x10, a, frames, s, 0);
            if (tw > 0 && (size_t)tw*th*48 <= VIS_LDS_LIMIT) {           // a window wider than the fixed tile: tile sized per launch
                RenderArgs d = a;
                d.tile_pitch = tw; d.tile_rows = th;
                return render_visualizer_tiled(TILED_R_DYNAMIC, d, frames, s, (size_t)tw*th*48);
            }
            return render_plain(fragment, a, frames, s);
        }
        case FRAG_MULTIPASS: {
            const int fast = launch_multipass_layer1(g_launch_ctx, a, frames, s);
            if (fast != 0) return fast < 0 ? fast : SFX_OK;
            return render_plain(fragment, a, frames, s);
        }
        case FRAG_MOTIONBLUR: {
            const int fast = launch_motionblur_layer1(g_launch_ctx, a, frames, s);
            if (fast != 0) return fast < 0 ? fast : SFX_OK;
            return render_plain(fragment, a, frames, s);
        }
        default: return render_plain(fragment, a, frames, s);
    }
}

#ifndef VIS_ROLLED_LANE_COST
#define VIS_ROLLED_LANE_COST 2.0f                                       // (see launch_fused: the shapes of rolled cameras)
#endif

#ifdef SF_SECTION_TIMERS
static int launch_fused_inner(int fragment, const RenderArgs& a, int ssaa, int frames, hipStream_t s, bool force_generic);
// profiling builds: run the launch with section timers and print the share of wave time per section
static int launch_fused(int fragment, const RenderArgs& a0, int ssaa, int frames, hipStream_t s, bool force_generic = false) {
    static unsigned long long* d_timers = nullptr;
    static std::vector<unsigned long long> host(SF_TIMER_ROWS*8);
    if (!d_timers) hipMalloc(&d_timers, host.size()*sizeof(unsigned long long));
    hipMemsetAsync(d_timers, 0, host.size()*sizeof(unsigned long long), s);
    RenderArgs a = a0;
    a.timers = d_timers;
    const int rc = launch_fused_inner(fragment, a, ssaa, frames, s, force_generic);
    unsigned long long t[8] = {};
    hipMemcpyAsync(host.data(), d_timers, host.size()*sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    for (size_t k = 0; k < host.size(); k++) t[k % 8] += host[k];
    if (g_last_kernel.rfind("k_visualizer_strip<", 0) == 0) {
        // the strip kernel's phases (visualizer_fast.hpp VisualizerStrip::run): wave cycles incl. stalls; column 7 counts diagonal folds
        const double total = (double)(t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6]);
        static const char* names[] = {"prologue+stage+barrier", "row-lines", "column-lines", "diagonals", "post", "exchange+resolve", "store"};
        const double waves = (double)frames*(double)((a.wr + 255)/256)*(double)((a.hr + 17)/18)*8.0;
        fprintf(stderr, "[section timers] %s, %d frames:", g_last_kernel.c_str(), frames);
        for (int k = 0; k < 7; k++) fprintf(stderr, " %s %.1f%%", names[k], 100.0*(double)t[k]/total);
        fprintf(stderr, " | %.0f wave-cycles per wave, %.2f diagonal folds per wave (of 180 advances)\n", total/waves, (double)t[7]/waves);
        return rc;
    }
    const double total = (double)(t[0] + t[1] + t[2] + t[3]);
    static const char* names[] = {"varyings+pre", "setup", "run", "resolve+store", "  setup.reduce", "  setup.stage", "  setup.barrier", "  run.blur"};
    fprintf(stderr, "[section timers] %d frames:", frames);
    for (int k = 0; k < 8; k++) fprintf(stderr, " %s %.1f%%", names[k], 100.0*(double)t[k]/total);
    fprintf(stderr, "\n");
    return rc;
}
#define launch_fused_body launch_fused_inner
#else
#define launch_fused_body launch_fused
#endif

static int launch_fused_body(int fragment, const RenderArgs& a, int ssaa, int frames, hipStream_t s, bool force_generic
#ifndef SF_SECTION_TIMERS
                             = false
#endif
                             ) {
    using namespace sfl;
    if (ssaa != 1 && ssaa != 2 && ssaa != 4) return fail(SFX_E_UNSUPPORTED, "fused ssaa %d", ssaa);
    switch (fragment) {
        case FRAG_DEFAULT:
            if (!force_generic) {
                const int fast = launch_separable(SEPARABLE_DEFAULT, g_launch_ctx, a, ssaa, frames, s);
                if (fast != 0) return fast < 0 ? fast : SFX_OK;
            }
            return fused_plain(fragment, a, ssaa, frames, s);
        case FRAG_VISUALIZER:
            if (!force_generic) {
                const int fast = launch_visualizer_fast(g_launch_ctx, a, ssaa, frames, s, false);
                if (fast != 0) return fast < 0 ? fast : SFX_OK;
            }
            if (!force_generic && visualizer_tile_applicable(a.tex[TEX_BACKGROUND])) {
                // the block shades 128*ssaa x rows*ssaa samples (S == 1: 128 x 2 pixels); pick the fixed tile when its window fits
                int tw = 0, th = 0, pitch_ss = 0, rows_ss = 0, block_px = 0, block_rows = 0;
                tiled_fused_limits(pitch_ss, rows_ss, block_px, block_rows);
                visualizer_window_bound(a, (ssaa == 1 ? 128 : block_px)*ssaa, (ssaa == 1 ? 2 : block_rows)*ssaa, tw, th);
                if (ssaa == 1 && tw <= 128 && th <= 10) return fused_visualizer_tiled(TILED_F_128x10, a, ssaa, frames, s, 0);
                if (ssaa != 1 && tw <= pitch_ss && th <= rows_ss) {
                    // four samples per lane need more registers: 6 waves per SIMD without spills beat 8 with (8K 4xSSAA: 55 -> 63 frames/s)
                    return fused_visualizer_tiled(ssaa == 4 ? TILED_F_SS_S4 : TILED_F_SS, a, ssaa, frames, s, 0);
                }
                if (ssaa == 2) {
                    // 0.43 texel per sample (1080p output at 2x SSAA over a 1080-row background): 64 pixels x 2 rows per block see a
                    // 63 x 10 window — a 64 x 11 tile (33 KB) keeps four 512-thread blocks on a CU, where the tile sized per launch
                    // below holds two
                    int tw2 = 0, th2 = 0;
                    visualizer_window_bound(a, 64*2, 2*2, tw2, th2);
                    if (tw2 <= 64 && th2 <= 11) return fused_visualizer_tiled(TILED_F_64x11, a, ssaa, frames, s, 0);
                }
                if (ssaa == 4) {
                    // the same for 4x SSAA at 0.2-0.3 texel per sample (1080p / 720p outputs): 32 pixels x 4 rows per block, a 56 x 14 tile (37 KB)
                    int tw4 = 0, th4 = 0;
                    visualizer_window_bound(a, 32*4, 4*4, tw4, th4);
                    if (tw4 <= 56 && th4 <= 14) return fused_visualizer_tiled(TILED_F_56x14, a, ssaa, frames, s, 0);
                }
                if (ssaa == 2 && !a.identity_camera && !a.axis_camera) {
                    // a rolled or tilted camera: a block's window grows with the block's extent along BOTH axes, so squarer blocks
                    // stage fewer cells per pixel (C3 rolled by 45 degrees: 64 x 2 pixels see 31 x 31 cells, 32 x 4 see 23 x 23)
                    // (round 5: and quads that WALK two or four rows — one sample per lane left every per-block cost, the camera's ray, the
                    // window and the staging, 56 % of a rolled launch, to be paid per sample)
                    struct Shape { int px, rows, walk; } const shapes[] = {{128, 1, 1}, {64, 2, 1}, {32, 4, 1}, {32, 4, 4}};
                    // (measured at C3, 17 / 45 degrees: 32 x 4 x 1: 684 / 662 frames/s, 32 x 4 x 4: 743 / 663; the two-row walks and the
                    // 64-wide ones spill or leave blocks off their tile and lose: profiles/r05_rolled_camera.txt)
                    int best = -1, best_tw = 0, best_th = 0;
                    float best_cost = 0.0f;
                    static const int only = [] { const char* e = getenv("SHADERFLOW_VIS_SHAPE"); return e ? atoi(e) : -1; }();   // A/B switch for measurements
                    for (int k = 0; k < 4; k++) {
                        int w = 0, h = 0;
                        if (only >= 0 && k != only) continue;
                        visualizer_window_bound(a, shapes[k].px*2, shapes[k].rows*shapes[k].walk*2, w, h);
                        // an odd pitch: a cell is 12 dwords, and with the camera turned by a quarter the lanes of a wave read down a
                        // COLUMN of cells — 16 cells per row put every one of them on the same banks (730 -> 448 frames/s at C3, 90 degrees).
                        // (made odd BEFORE the LDS test: what is checked is what is launched)
                        w |= 1;
                        if ((size_t)w*h*48 > 72*1024) continue;                       // two 512-thread blocks per CU at least
                        // cells staged per pixel + what a block pays once per LANE (ray set-up, window, barriers), in cells' worth
                        const float cost = (float)w*(float)h/(float)(shapes[k].px*shapes[k].rows*shapes[k].walk) + VIS_ROLLED_LANE_COST/(float)shapes[k].walk;
                        if (best < 0 || cost < best_cost) { best = k; best_cost = cost; best_tw = w; best_th = h; }
                    }
                    if (best >= 0) {
                        RenderArgs d = a;
                        d.tile_pitch = best_tw; d.tile_rows = best_th;
                        const size_t lds = (size_t)best_tw*best_th*48;
                        // (best == 1, 2: eight waves per SIMD, not the four of the other tiles sized per launch: 555 -> 696 frames/s at C3 rolled by 17 degrees)
                        static const TiledFused kernels[] = {TILED_F_DYN_128, TILED_F_DYN_64x2, TILED_F_DYN_32x4, TILED_F_DYN_32x4_WALK4};
                        return fused_visualizer_tiled(kernels[best], d, ssaa, frames, s, lds);
                    }
                }
                if (ssaa != 1) {
                    // denser backgrounds (1080p output at 2x SSAA over a 1080-row background: 0.43 texel per sample; backgrounds
                    // larger than the output): the tile is sized per launch in dynamic LDS, and the block narrows from 128 to 64
                    // or 32 pixels until its window leaves room for at least two blocks per CU
                    const int rows = block_rows*ssaa;
                    int best_px = 0, best_tw = 0, best_th = 0;
                    for (int px : {128, 64, 32}) {
                        visualizer_window_bound(a, px*ssaa, rows, tw, th);
                        tw |= 1;                                                      // an odd pitch (see above), before the LDS test
                        const size_t lds = (size_t)tw*th*48;
                        if (lds <= VIS_LDS_LIMIT) { best_px = px; best_tw = tw; best_th = th; if (lds <= 64*1024) break; }
                    }
                    if (best_px) {
                        RenderArgs d = a;
                        d.tile_pitch = best_tw; d.tile_rows = best_th;
                        const size_t lds = (size_t)best_tw*best_th*48;
                        return fused_visualizer_tiled(best_px == 128 ? TILED_F_DYN_128 : (best_px == 64 ? TILED_F_DYN_64 : TILED_F_DYN_32), d, ssaa, frames, s, lds);
                    }
                } else if ((size_t)tw*th*48 <= VIS_LDS_LIMIT) {
                    RenderArgs d = a;
                    d.tile_pitch = tw; d.tile_rows = th;
                    return fused_visualizer_tiled(TILED_F_DYN_1X, d, ssaa, frames, s, (size_t)tw*th*48);
                }
            }
            return fused_plain(fragment, a, ssaa, frames, s);
        case FRAG_BARS: {
            const int fast = force_generic ? 0 : launch_separable(SEPARABLE_BARS, g_launch_ctx, a, ssaa, frames, s);
            if (fast != 0) return fast < 0 ? fast : SFX_OK;
            return fused_plain(fragment, a, ssaa, frames, s);
        }
        case FRAG_WAVEFORM: {
            const int fast = force_generic ? 0 : launch_separable(SEPARABLE_WAVEFORM, g_launch_ctx, a, ssaa, frames, s);
            if (fast != 0) return fast < 0 ? fast : SFX_OK;
            return fused_plain(fragment, a, ssaa, frames, s);
        }
        default: return fused_plain(fragment, a, ssaa, frames, s);
    }
}

// A loaded program runs the generic kernels of its own code object (PlainShader geometry, jit_runtime.hpp)
static int launch_jit(hipFunction_t fn, const RenderArgs& a, dim3 grid, dim3 block, hipStream_t s) {
    size_t size = sizeof(RenderArgs);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, (void*)&a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    const hipError_t e = hipModuleLaunchKernel(fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, 0, s, nullptr, config);
    return e == hipSuccess ? SFX_OK : fail(SFX_E_HIP, "hipModuleLaunchKernel: %s", hipGetErrorString(e));
}
static int launch_render_p(const Program* p, const RenderArgs& a, int frames, hipStream_t s) {
    g_launch_ctx = p->ctx;
    if (p->fragment != FRAG_JIT) return launch_render(p->fragment, a, frames, s);
    using P = PlainShader<FRAG_DEFAULT>;
    if (a.quads && jit_rows_1x(p) > 1) {
        // a mipmapped sampler on a TILED translated fragment: its lanes walk rows and cannot form quads — the differences across a "quad"
        // would be between unrelated pixels and the level of detail wrong without any error. The code object's untiled twin takes the draw.
        if (!p->fn_render_quads) return fail(SFX_E_UNSUPPORTED, "a mipmapped texture is bound to a tiled translated fragment whose code object has no sfx_jit_render_quads: translate it again with this library's headers");
        return launch_jit(p->fn_render_quads, a, dim3((a.wr + P::BLOCK_W - 1)/P::BLOCK_W, (a.hr + P::BLOCK_H - 1)/P::BLOCK_H, frames), dim3(P::BLOCK_W, P::BLOCK_H, 1), s);
    }
    const int block_rows = P::BLOCK_H*jit_rows_1x(p);
    return launch_jit(p->fn_render, a, dim3((a.wr + P::BLOCK_W - 1)/P::BLOCK_W, (a.hr + block_rows - 1)/block_rows, frames), dim3(P::BLOCK_W, P::BLOCK_H, 1), s);
}
static int launch_fused_p(const Program* p, const RenderArgs& a, int ssaa, int frames, hipStream_t s) {
    g_launch_ctx = p->ctx;
    if (p->fragment != FRAG_JIT) return launch_fused(p->fragment, a, ssaa, frames, s);
    using P = PlainShader<FRAG_DEFAULT>;
    if (ssaa == 1) { const int rows = 2*jit_rows_1x(p); return launch_jit(p->fn_fused[0], a, dim3(((a.w + 127)/128)*((a.h + rows - 1)/rows), 1, frames), dim3(256), s); }
    if (ssaa != 2 && ssaa != 4) return fail(SFX_E_UNSUPPORTED, "fused ssaa %d", ssaa);
    constexpr int rows = P::FUSED_ROWS*P::THREAD_ROWS, threads = 4*P::BLOCK_PX*P::THREAD_ROWS;
    const int blocks_x = (a.w + P::BLOCK_PX - 1)/P::BLOCK_PX, row_blocks = (a.h + rows - 1)/rows;
    return launch_jit(p->fn_fused[ssaa == 2 ? 1 : 2], a, dim3(blocks_x*row_blocks, 1, frames), dim3(threads), s);
}

extern "C" int sfx_render(sfx_handle h, sfx_handle target, int layer) {
    Program* p = get<Program>(h, MAGIC_PROG);
    Texture* t = get<Texture>(target, MAGIC_TEX);
    if (!p || !t) return fail(SFX_E_INVALID, "invalid program or target handle");
    if (p->fragment == FRAG_FINAL) return fail(SFX_E_INVALID, "the final program is driven by sfx_resolve / sfx_render_resolve");
    if (t->dtype != SFX_U8 && t->dtype != SFX_F32 && t->dtype != SFX_F16) return fail(SFX_E_UNSUPPORTED, "render target dtype %d", t->dtype);
    USE_DEVICE(p->ctx);
    RenderArgs a;
    fill_args(p, a);
    a.u.iLayer = layer;                                             // shader.py:402
    a.wr = t->width; a.hr = t->height; a.w = t->width; a.h = t->height;
    a.out = t->data; a.out_components = t->components; a.out_dtype = t->dtype; a.out_frame_stride = 0;
    set_pixel_centres(a);
    int rc = check_samplers(p->fragment, a);
    if (rc) return rc;
    if ((rc = launch_render_p(p, a, 1, p->ctx->stream))) return rc;
    return launch_status();
}


extern "C" int sfx_resolve(sfx_handle h, sfx_handle src, sfx_handle dst, int subsample) {
    CTX_OR_FAIL(c, h);
    Texture* s = get<Texture>(src, MAGIC_TEX);
    Texture* d = get<Texture>(dst, MAGIC_TEX);
    if (!s || !d) return fail(SFX_E_INVALID, "invalid texture handle");
    if (s->dtype != SFX_U8 || s->components != 4) return fail(SFX_E_UNSUPPORTED, "resolve source must be RGBA8 (iScreen)");
    if (d->dtype != SFX_U8 || d->components != 3) return fail(SFX_E_UNSUPPORTED, "resolve target must be RGB8 (iFinal, scene.py:188-189)");
    USE_DEVICE(c);
    ResolveArgs a;
    a.screen = tex_view(s);
    a.screen.repeat_x = s->repeat_x; a.screen.repeat_y = s->repeat_y;
    a.w = d->width; a.h = d->height; a.subsample = subsample < 1 ? 1 : subsample;
    a.out = (uint8_t*)d->data;
    a.screen_frame_stride = 0; a.out_frame_stride = 0; a.top_down = c->top_down;
    { const int rc = sfl::launch_resolve(c, a, 1, c->stream); if (rc) return rc; }
    return launch_status();
}

extern "C" int sfx_fused_supported(int ssaa_x1000, int subsample) {
    if (ssaa_x1000 % 1000) return 0;
    return fused_supported(ssaa_x1000/1000, subsample < 1 ? 1 : subsample) ? 1 : 0;
}

extern "C" int sfx_render_resolve(sfx_handle h, sfx_handle final_tex, int ssaa, int subsample) {
    Program* p = get<Program>(h, MAGIC_PROG);
    Texture* t = get<Texture>(final_tex, MAGIC_TEX);
    if (!p || !t) return fail(SFX_E_INVALID, "invalid program or target handle");
    if (t->dtype != SFX_U8 || t->components != 3) return fail(SFX_E_UNSUPPORTED, "fused target must be RGB8 (iFinal, scene.py:188-189)");
    if (subsample < 1) subsample = 1;
    if (!fused_supported(ssaa, subsample)) return fail(SFX_E_UNSUPPORTED, "final.glsl footprint for ssaa=%d subsample=%d leaves the pixel's block: use sfx_render + sfx_resolve", ssaa, subsample);
    if (p->ctx->filter_model != SFX_FILTER_SPEC) return fail(SFX_E_UNSUPPORTED, "the context's fixed-point filter model filters final.glsl's taps as well: use sfx_render + sfx_resolve");
    if (!fusable(p, ssaa)) return fail(SFX_E_UNSUPPORTED, "the fragment takes screen-space derivatives, which the fused kernel's lane layout provides for ssaa 2 only: use sfx_render + sfx_resolve");
    USE_DEVICE(p->ctx);
    RenderArgs a;
    fill_args(p, a);
    a.w = t->width; a.h = t->height; a.wr = t->width*ssaa; a.hr = t->height*ssaa; a.subsample = subsample;
    a.out = t->data; a.out_frame_stride = 0;
    set_pixel_centres(a);
    int rc = check_samplers(p->fragment, a);
    if (rc) return rc;
    if ((rc = launch_fused_p(p, a, ssaa, 1, p->ctx->stream))) return rc;
    return launch_status();
}


// ---------------------------------------------------------------------------------------------------------
// Clock sequence: the frame loop of scenes in which nothing but the clock moves, WITHOUT a host language between the frames.
// scene.next (scene.py:456-479) → every program's render (shader.py:388-405: a draw per layer into row 0 of its texture matrix, then the
// matrix rolls, texture.py:295-298) → iFinal's resolve → exporting.pipe (exporting.py:151-174), `nframes` times in one call. Per frame
// the host side is: four stores per program (the clock uniforms), one pointer per sampler of a rolled matrix, the launches, one ring
// hand-off — a few microseconds, where clockloop.ClockLoop's python needed ~70. The same launches with the same arguments in the same
// order as ShaderScene.next, so the same frames (tests/test_gpu_multipass.py::test_clock_sequence_gives_the_frame_loops_bytes).
static int resolve_sampler(const Program* p, const char* name) {
    if (p->fragment == FRAG_JIT) { for (const auto& b : p->bindings) if (b.sampler && b.name == name) return b.slot; return -1; }
    return sampler_slot(p->fragment, name);
}
extern "C" int sfx_clock_sequence_run(sfx_handle hc, const sfx_sequence_pass* passes, int npasses, const sfx_sequence_matrix* matrices, int nmatrices,
                                      const sfx_clock_tick* clock, int nframes, sfx_handle hring, int first_slot, int fd,
                                      void* const* planar_slots, int yuv_matrix, int width, int height) {
    CTX_OR_FAIL(c, hc);
    const int ring_slots = hring ? ring_slot_count(hring) : 0;
    if (!passes || npasses < 1 || nmatrices < 0 || (nmatrices && !matrices) || !clock || nframes < 0) return fail(SFX_E_INVALID, "clock sequence: null tables");
    if (ring_slots < 0) return fail(SFX_E_INVALID, "clock sequence: invalid ring handle");
    USE_DEVICE(c);
    // the matrices as this call rolls them: order[m][t] = row of the caller's table that sits at depth t now
    std::vector<std::vector<int>> order(nmatrices);
    for (int m = 0; m < nmatrices; m++) {
        if (matrices[m].temporal < 1 || matrices[m].layers < 1 || !matrices[m].textures) return fail(SFX_E_INVALID, "clock sequence: matrix %d", m);
        order[m].resize(matrices[m].temporal);
        for (int t = 0; t < matrices[m].temporal; t++) order[m][t] = t;
        for (int k = 0; k < matrices[m].temporal*matrices[m].layers; k++) {
            Texture* texture = get<Texture>(matrices[m].textures[k], MAGIC_TEX);
            if (!texture || texture->ctx != c) return fail(SFX_E_INVALID, "clock sequence: matrix %d, box %d is not a texture of this context", m, k);
        }
    }
    auto box = [&](int m, int t, int l) -> sfx_handle { return matrices[m].textures[order[m][t]*matrices[m].layers + (l < 0 ? matrices[m].layers + l : l)]; };
    // sampler slots of every (program, named box), resolved once: the names never change, only what sits behind them
    struct Bind { Program* p; int slot, m, t, l; };
    std::vector<Bind> binds;
    for (int k = 0; k < npasses; k++) {
        if (passes[k].kind == SFX_PASS_RESOLVE) continue;
        Program* p = get<Program>(passes[k].program, MAGIC_PROG);
        if (!p || p->ctx != c) return fail(SFX_E_INVALID, "clock sequence: pass %d has no program of this context", k);
        if (passes[k].matrix < 0 || passes[k].matrix >= nmatrices) return fail(SFX_E_INVALID, "clock sequence: pass %d names matrix %d", k, passes[k].matrix);
        for (int m = 0; m < nmatrices; m++) {
            if (!matrices[m].names || matrices[m].temporal < 2) continue;     // (samplers of a matrix that never rolls were bound by the host)
            for (int t = 0; t < matrices[m].temporal; t++) for (int l = 0; l < matrices[m].layers; l++) {
                const char* name = matrices[m].names[t*matrices[m].layers + l];
                const int slot = name ? resolve_sampler(p, name) : -1;
                if (slot >= 0) binds.push_back({p, slot, m, t, l});
            }
        }
    }
    for (int f = 0; f < nframes; f++) {
        const sfx_clock_tick& now = clock[f];
        bool fused = false;
        for (int k = 0; k < npasses; k++) {
            const sfx_sequence_pass& pass = passes[k];
            if (pass.kind == SFX_PASS_RESOLVE) {
                if (fused) continue;                                // the main pass resolved into iFinal already (shader.py:391-396)
                if (pass.matrix < 0 || pass.matrix >= nmatrices) return fail(SFX_E_INVALID, "clock sequence: resolve pass %d names matrix %d", k, pass.matrix);
                const int rc = sfx_resolve(hc, box(pass.matrix, 0, -1), pass.target, pass.subsample);
                if (rc) return rc;
                continue;
            }
            Program* p = get<Program>(pass.program, MAGIC_PROG);
            p->u.iTime = now.time; p->u.iTau = now.tau; p->u.iDeltatime = now.deltatime; p->u.iFrame = now.frame;      // sfx_uniform_set_clock
            for (const Bind& b : binds) if (b.p == p) b.p->samplers[b.slot] = get<Texture>(box(b.m, b.t, b.l), MAGIC_TEX);
            if (pass.kind == SFX_PASS_FUSED) {
                const int rc = sfx_render_resolve(pass.program, pass.target, pass.ssaa, pass.subsample);
                if (rc) return rc;
                fused = true;
            } else {
                for (int l = 0; l < matrices[pass.matrix].layers; l++) {
                    const int rc = sfx_render(pass.program, box(pass.matrix, 0, l), l);      // shader.py:400-403: iLayer = l, into row 0
                    if (rc) return rc;
                }
            }
            std::vector<int>& rows = order[pass.matrix];              // texture.roll(): the oldest row becomes row 0
            std::rotate(rows.begin(), rows.end() - 1, rows.end());
        }
        if (ring_slots > 0 && fd >= 0) {                            // exporting.pipe (exporting.py:151-174)
            const int slot = (first_slot + f) % ring_slots;
            int rc;
            if (planar_slots) {
                if ((rc = sfx_ring_pipe_sync(hring, slot))) return rc;
                Texture* final_texture = get<Texture>(passes[npasses - 1].target, MAGIC_TEX);
                if (!final_texture) return fail(SFX_E_INVALID, "clock sequence: the last pass has no target to convert");
                // (the conversion reads width x height RGB8 texels and writes width*height*3/2 bytes: the caller's numbers must be the texture's)
                if (final_texture->dtype != SFX_U8 || final_texture->components != 3 || final_texture->width != width || final_texture->height != height)
                    return fail(SFX_E_INVALID, "clock sequence: yuv420p of a %d x %d frame was asked for, the last pass' target is %d x %d x %d (dtype %d)",
                                width, height, final_texture->width, final_texture->height, final_texture->components, final_texture->dtype);
                if ((rc = sfx_rgb_to_yuv420(hc, final_texture->data, planar_slots[slot], width, height, 1, yuv_matrix))) return rc;
                rc = sfx_ring_read_device_async(hring, planar_slots[slot], slot);
            } else {
                rc = sfx_ring_read_async(hring, passes[npasses - 1].target, slot);
            }
            if (rc) return rc;
            if ((rc = sfx_ring_pipe(hring, slot, fd))) return rc;
        }
    }
    return SFX_OK;
}


extern "C" int sfx_render_tape(sfx_handle hp, sfx_handle ht, int frame0, int nframes, int width, int height,
                               int ssaa_x1000, int subsample, void* device_out) {
    Program* p = get<Program>(hp, MAGIC_PROG);
    TapeView tape, *t = &tape;                                      // the bank the last build filled (capi_audio.hip)
    if (!p || !tape_view(ht, t) || !device_out) return fail(SFX_E_INVALID, "invalid program/tape handle or output");
    if (p->ctx != t->ctx) return fail(SFX_E_INVALID, "program and tape belong to different contexts");
    if (frame0 < 0 || nframes < 1 || frame0 + nframes > t->max_frames) return fail(SFX_E_INVALID, "frames [%d, %d) outside the tape", frame0, frame0 + nframes);
    if (subsample < 1) subsample = 1;
    if (ssaa_x1000 < 10) return fail(SFX_E_INVALID, "ssaa %d/1000", ssaa_x1000);
    const bool fused = (ssaa_x1000 % 1000 == 0) && fused_supported(ssaa_x1000/1000, subsample) && fusable(p, ssaa_x1000/1000);
    const int ssaa = ssaa_x1000/1000;
    USE_DEVICE(p->ctx);
    RenderArgs a;
    fill_args(p, a);
    a.w = width; a.h = height; a.subsample = subsample;
    a.wr = (int)((double)width*ssaa_x1000/1000.0); a.hr = (int)((double)height*ssaa_x1000/1000.0);   // scene.py:372-375
    set_pixel_centres(a);
    a.out = device_out; a.out_frame_stride = (long)width*height*3;
    a.dyn = t->dyn; a.frame0 = frame0;
    // the bank was filled on the tape's stream: this stream waits for its last audio kernel, and leaves a mark behind its own last
    // kernel that the build refilling the bank will wait for (Tape, above)
    HIP_TRY(hipStreamWaitEvent(p->ctx->stream, t->built, 0));
    struct RenderedMark { hipEvent_t event; hipStream_t stream; ~RenderedMark() { hipEventRecord(event, stream); } } mark{t->rendered, p->ctx->stream};
    if (t->audio) {
    // iSpectrogram: width 1 (length=0 scenes), height bins, RG32F (spectrogram.py:298-311); the bound texture's
    // sampler state is kept, only its storage is redirected to the tape column of the frame
    // (length > 0: the texture is `width` columns wide and every frame of the batch gets its own state of it, k_spectrogram_scroll)
    const bool scrolling = (t->width > 1);
    a.tape_spectrogram = scrolling ? t->scroll : t->columns; a.spectrogram_stride = (long)t->values*t->width;
    if (!a.tex[TEX_SPECTROGRAM].data) {
        Tex& s = a.tex[TEX_SPECTROGRAM];
        s.width = t->width; s.height = t->bins; s.components = t->channels; s.dtype = DT_F32; s.filter = FILTER_NEAREST; s.repeat_x = 1; s.repeat_y = 0;
    }
    if (a.tex[TEX_SPECTROGRAM].width != t->width || a.tex[TEX_SPECTROGRAM].height != t->bins || a.tex[TEX_SPECTROGRAM].components != t->channels)
        return fail(SFX_E_INVALID, "the bound iSpectrogram is %d x %d x %d, the tape was created for %d x %d x %d (length_samples x bins x channels)",
                    a.tex[TEX_SPECTROGRAM].width, a.tex[TEX_SPECTROGRAM].height, a.tex[TEX_SPECTROGRAM].components, t->width, t->bins, t->channels);
    if (t->points > 0) {
        a.tape_waveform = t->rows; a.waveform_stride = (long)t->points*t->pcm_channels;
        if (!a.tex[TEX_WAVEFORM].data) {
            Tex& w = a.tex[TEX_WAVEFORM];
            w.width = t->points; w.height = 1; w.components = t->pcm_channels; w.dtype = DT_F32; w.filter = FILTER_LINEAR; w.repeat_x = 0; w.repeat_y = 0;
        }
    }
    // placeholders so that texel() sees a non-null base before frame_view() redirects it
    a.tex[TEX_SPECTROGRAM].data = a.tape_spectrogram;
    if (t->points > 0) a.tex[TEX_WAVEFORM].data = t->rows;
    }
    int rc = check_samplers(p->fragment, a);
    if (rc) return rc;
    a.has_vis = 0;                                                  // per-frame audio uniforms live on the device
    if (p->fragment == FRAG_VISUALIZER) {
        sfl::visualizer_consts_frames(t->dyn, frame0, nframes, t->vis, p->ctx->stream);
        a.vis_consts = t->vis;
        if (t->audio && t->width == 1 && a.tex[TEX_SPECTROGRAM].components == 2 && a.tex[TEX_SPECTROGRAM].filter == FILTER_NEAREST) {
            const long count = (long)nframes*t->values;
            sfl::visualizer_bars(t->columns + (long)frame0*t->values, count, t->bars + (long)frame0*t->values, p->ctx->stream);
            a.tape_bars = t->bars;
        }
    }
    if (fused) {
        if ((rc = launch_fused_p(p, a, ssaa, nframes, p->ctx->stream))) return rc;
        return launch_status();
    }
    // two passes, batched: the fragment into an RGBA8 iScreen scratch per frame, then final.glsl (shader.py:388-405)
    const size_t screen_frame = (size_t)a.wr*a.hr*4;
    void* screen = nullptr;
    if ((rc = tape_screen_scratch(ht, screen_frame*nframes, p->ctx->stream, &screen))) return rc;
    a.out = screen; a.out_frame_stride = (long)screen_frame; a.out_components = 4; a.out_dtype = DT_U8;
    if ((rc = launch_render_p(p, a, nframes, p->ctx->stream))) return rc;
    ResolveArgs r;
    r.screen = Tex{screen, a.wr, a.hr, 4, DT_U8, p->ctx->filter_model == SFX_FILTER_FIXED8 ? FILTER_LINEAR_FIXED8 : FILTER_LINEAR, 0, 0};      // iScreen: linear, repeat(False) (scene.py:192-194)
    r.w = width; r.h = height; r.subsample = subsample; r.out = (uint8_t*)device_out;
    r.screen_frame_stride = (long)screen_frame; r.out_frame_stride = (long)width*height*3; r.top_down = p->ctx->top_down;
    if ((rc = sfl::launch_resolve(p->ctx, r, nframes, p->ctx->stream))) return rc;
    return launch_status();
}
