// host_state.hpp — what the translation units of libshaderflow_hip.so share on the HOST side: error reporting, handles, the context and
// texture objects, and the few calls the C-ABI units (capi.hip, capi_readout.hip, capi_audio.hip) make into each other. capi.hip owns the
// definitions of the thread-local state; the launch units (launch_*.hip) only read it.
#pragma once

#include "../../include/shaderflow_hip.h"
#include "render_kernels.hpp"

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern thread_local std::string g_last_kernel;   // which render kernel instance the last launch on this thread picked (sfx_last_kernel)
int fail(int code, const char* fmt, ...);         // sets sfx_last_error() of this thread, returns `code`
int launch_status();                              // hipGetLastError() after a launch as an sfx status (capi.hip)
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(SFX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

enum : uint32_t { MAGIC_CTX = 0x53465843, MAGIC_TEX = 0x53465854, MAGIC_PROG = 0x53465850, MAGIC_RING = 0x53465852,
                  MAGIC_AUDIO = 0x53465841, MAGIC_PLAN = 0x5346584c, MAGIC_TAPE = 0x53465854 + 0x100, MAGIC_SHM = 0x53465853 };

struct Object { uint32_t magic; };

template <class T> static T* get(sfx_handle h, uint32_t magic) {
    Object* o = reinterpret_cast<Object*>(static_cast<uintptr_t>(h));
    return (o && o->magic == magic) ? static_cast<T*>(o) : nullptr;
}
template <class T> static sfx_handle handle_of(T* p) { return static_cast<sfx_handle>(reinterpret_cast<uintptr_t>(p)); }

namespace sf { struct MultipassTaps; }
struct Context : Object {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t events[64] = {};
    hipDeviceProp_t prop;
    float tap_x[81], tap_y[81];
    std::vector<struct Program*> programs;   // live programs of this context (their sampler slots point at textures)
    int top_down = 0;                // frames leave with rows top-down (sfx_ctx_output_top_down)
    int filter_model = SFX_FILTER_SPEC;   // sfx_ctx_filter_model: how LINEAR unorm8 textures of this context are filtered (tex_view)
    // per-frame column/row tables of the fast visualizer kernel (visualizer_fast.hpp), grown on demand
    void* vis_tables = nullptr; size_t vis_tables_bytes = 0;
    float* vis_bars = nullptr; size_t vis_bars_count = 0;          // sqrt(texel/1000) of a bound spectrogram (single launches)
    void* resolve_tables = nullptr; size_t resolve_tables_bytes = 0;   // column/row tap tables of k_resolve_fast
    long resolve_tables_key[8] = {};                               // the geometry (and stream) they were built for
    // the context's two copy streams (read-out ring, shared-memory ring, peer windows): chosen once so that neither shares a hardware
    // queue with `stream` (context_copy_streams)
    hipStream_t copy_streams[2] = {nullptr, nullptr};
    unsigned* tile_misses = nullptr;                               // device counter of sfx_ctx_tile_misses, allocated by its first call
    struct sf::MultipassTaps* multipass_taps = nullptr;            // multipass.frag's blur taps on the device (layered_fast.hpp), built on first use
    float multipass_reach[2] = {0.0f, 0.0f};
    int copy_candidates = 0, copy_colliding = 0;                   // how many streams the choice looked at / found serialised behind `stream`
    // read-out rings of this context: (ring, "every frame handed to it so far has left device memory"). Their copies run outside HIP's
    // queues, so hipFree's implicit wait knows nothing of them: sfx_device_free asks them first.
    std::vector<std::pair<void*, void (*)(void*)>> readouts;
    struct EngineCopy* engines = nullptr;                           // agents and SDMA engines of the read-out (EngineLanes); null until first use
    struct EngineCopy* peer_engines = nullptr;                      // … of the peer copies (PeerCopier)
    struct PeerCopier* peer = nullptr;                              // the sharded export's peer copies: a thread that issues them on named engines
    // peer copies of the sharded export's "device-sdma" mode: the copy streams, an event per lane (sfx_peer_*)
};
extern thread_local Context* g_launch_ctx;               // the context whose program is being launched (scratch owner)

struct Texture : Object {
    Context* ctx;
    int width, height, components, dtype, filter = SFX_LINEAR, repeat_x = 1, repeat_y = 1;
    void* data = nullptr;
    size_t nbytes = 0;
    void* mips = nullptr; int levels = 1;      // levels 1… of the chain, built by sfx_texture_build_mipmaps (glsl.hpp mip_level says where each one starts)
};

#define CTX_OR_FAIL(var, h) Context* var = get<Context>(h, MAGIC_CTX); if (!var) return fail(SFX_E_INVALID, "invalid context handle")
#define USE_DEVICE(ctx) HIP_TRY(hipSetDevice((ctx)->device))

// ---- between the C-ABI units ------------------------------------------------------------------------------------------------------
// capi_readout.hip
void readout_release(Context* c);                 // sfx_ctx_destroy: the context's peer copier, copy streams and engine records
int ring_slot_count(sfx_handle ring);             // slots of a read-out ring; -1: not a ring
// capi_audio.hip: what a render reads of a tape — the bank its last build filled (sfx_render_tape)
struct TapeView {
    Context* ctx; int max_frames;
    bool audio;                                   // false: a clock tape (per-frame uniforms only)
    sf::FrameDyn* dyn; sf::VisualizerConsts* vis;
    hipEvent_t built, rendered;                   // the bank's: renders wait for `built`, and leave `rendered` behind their last kernel
    int width, values, bins, channels;            // iSpectrogram: length_samples x bins x channels (values = bins*channels)
    int points, pcm_channels;                     // iWaveform (points 0: none)
    float *columns, *scroll, *rows, *bars;
};
bool tape_view(sfx_handle tape, TapeView* view);
int tape_screen_scratch(sfx_handle tape, size_t bytes, hipStream_t stream, void** screen);   // iScreen of the two-pass path, grown on demand
