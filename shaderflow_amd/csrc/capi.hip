// capi.hip — implementation of include/shaderflow_hip.h (the C-ABI over the gfx950 kernels).
// Host-side state only: handles, the fragment registry, uniform/sampler tables, launch geometry, the
// pinned read-out ring with its pipe writer thread, and the audio plan/tape objects.

#include "launch.hpp"
#include "launch_geometry.hpp"
#include "audio_kernels.hpp"
#include "visualizer_kernels.hpp"
#include "uniform_table.hpp"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <unistd.h>
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

static void peer_stop(Context* c);                                 // (defined with the peer windows)
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

// Copy streams (read-out ring, shared-memory ring, peer windows). HIP multiplexes its streams onto a handful of IN-ORDER hardware
// queues (GPU_MAX_HW_QUEUES, 4 by default; a new stream joins the queue with the fewest streams). A copy stream that lands on the
// render stream's queue puts its wait-for-the-copy barrier packets in front of the next render kernel: read-out and render stop
// overlapping — measured at C3: 2 080 → 1 215 frames/s (= render + copy in series), which is what happened whenever other streams
// had been created before (torch's pool of 32, an earlier export's; with GPU_MAX_HW_QUEUES=2 always) and explains the 1 770-1 860 of
// bench.py's export leg against 2 080 for the same export in a fresh process (profiles/r04_export_streams.txt). Stream priorities are
// no way out (queues of another priority: 830 frames/s). So the choice is MEASURED, once per context: candidates are created until two
// are found whose copies complete WHILE a kernel occupies the render stream.
__global__ void k_hold_stream(const int* release, long long ticks) {
    const long long start = wall_clock64();                         // 100 MHz: `ticks` bounds the hold whatever the host does
    while (__hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && wall_clock64() - start < ticks) __builtin_amdgcn_s_sleep(32);
}

static int context_copy_streams(Context* c) {
    if (c->copy_streams[0]) return SFX_OK;
    const char* off = getenv("SHADERFLOW_COPY_STREAM_PROBE");
    int* release = nullptr; void* pinned = nullptr; void* device = nullptr; hipEvent_t landed = nullptr;
    bool probe = !(off && !strcmp(off, "0"));
    if (probe && (hipHostMalloc((void**)&release, 4096, hipHostMallocMapped) != hipSuccess || hipHostMalloc(&pinned, 4096, hipHostMallocDefault) != hipSuccess ||
                  hipMalloc(&device, 4096) != hipSuccess || hipEventCreateWithFlags(&landed, hipEventDisableTiming) != hipSuccess)) { (void)hipGetLastError(); probe = false; }
    std::vector<hipStream_t> rejected;
    int found = 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int candidate = 0; found < 2 && candidate < 12; candidate++) {
        hipStream_t stream = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        c->copy_candidates++;
        bool independent = true;
        if (probe) {
            *release = 0;
            hipLaunchKernelGGL(k_hold_stream, dim3(1), dim3(1), 0, c->stream, release, 2000000LL);      // ≤ 20 ms, normally ≈ 0.1 ms
            hipMemcpyAsync(pinned, device, 4096, hipMemcpyDeviceToHost, stream);
            hipEventRecord(landed, stream);
            const auto started = std::chrono::steady_clock::now();
            independent = false;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - started).count() < 3e-3)
                if (hipEventQuery(landed) == hipSuccess) { independent = true; break; }
            (void)hipGetLastError();                                // hipErrorNotReady of the queries
            __atomic_store_n(release, 1, __ATOMIC_RELEASE);
            hipStreamSynchronize(c->stream);
            hipStreamSynchronize(stream);
        }
        if (independent) c->copy_streams[found++] = stream;
        else { c->copy_colliding++; rejected.push_back(stream); }
    }
    // nothing independent to be had (one hardware queue): the rejected ones still work, in series with the render
    while (found < 2 && !rejected.empty()) { c->copy_streams[found++] = rejected.back(); rejected.pop_back(); }
    for (hipStream_t stream : rejected) hipStreamDestroy(stream);
    if (release) hipHostFree(release);
    if (pinned) hipHostFree(pinned);
    if (device) hipFree(device);
    if (landed) hipEventDestroy(landed);
    if (found < 2) return fail(SFX_E_HIP, "no copy streams");
    return SFX_OK;
}

// Frames leave device memory through `EngineLanes`: two device-to-host copies in flight, each on an SDMA engine NAMED by this library.
//
// Not hipMemcpyAsync on a copy stream (rounds 1-3): the runtime picks an engine per stream — the lowest one free at that moment, then
// sticky — and the sixteen engines of an MI355X are far from equal for device-to-host traffic (tools/ubench_sdma_engines.hip,
// profiles/r04_export_streams.txt): engines 0-3 move 42-54 GB/s, 4-7 ≈ 12, 8-11 ≈ 9, 12-15 ≈ 7. A stream that draws a far engine reads
// 4K frames out at a quarter of the bus for the rest of its life: the 830 frames/s exports of the third context of a process. And a
// copy stream that shares a hardware queue with the render stream serialises read-out and render (1 215 frames/s). So the read-out
// owns no stream at all: a thread of the ring waits for the frame on the host (hipEventSynchronize of an event recorded on the render
// stream), hands it to HSA's copy-on-engine call on one of the two engines HSA itself recommends for this direction
// (hsa_amd_memory_get_preferred_copy_engine; SHADERFLOW_SDMA_ENGINES=a,b overrides) and waits for HSA's completion signal. Nothing
// of it passes through a HIP queue, so nothing of it depends on how the runtime folds streams onto queues. (A host FUNCTION on a copy
// stream doing the same was measured first: ≈ 1 ms of latency per callback — fine behind a deep queue, 800 frames/s in the frame loop.)
// When HSA does not answer (or SHADERFLOW_READOUT=hip) a lane is hipMemcpyAsync + hipStreamSynchronize on one of the context's
// probed copy streams.
struct EngineCopy {
    bool usable = false;
    hsa_agent_t gpu{}, cpu{};                                       // the source's agent (this context's GPU) and the destination's (a CPU socket, or — peer copies — the GPU that owns the window)
    uint32_t engine[2] = {0, 0};
};

static hsa_status_t collect_agents(hsa_agent_t agent, void* data) {
    auto* lists = (std::pair<std::vector<hsa_agent_t>, std::vector<hsa_agent_t>>*)data;
    hsa_device_type_t type;
    if (hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) == HSA_STATUS_SUCCESS) (type == HSA_DEVICE_TYPE_GPU ? lists->first : lists->second).push_back(agent);
    return HSA_STATUS_SUCCESS;
}

// the agents of a frame's two ends, from the pointers themselves; engines from HSA's recommendation. Once per context.
static EngineCopy* engine_copy(Context* c, const void* host, const void* device) {
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    if (c->engines) return c->engines->usable ? c->engines : nullptr;
    EngineCopy* e = c->engines = new EngineCopy();
    const char* route = getenv("SHADERFLOW_READOUT");
    if (route && strcmp(route, "engine")) return nullptr;
    if (hsa_init() != HSA_STATUS_SUCCESS) return nullptr;             // reference-counted: HIP holds the runtime open already
    hsa_amd_pointer_info_t info{}; info.size = sizeof(info);
    if (hsa_amd_pointer_info(device, &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || info.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return nullptr;
    e->gpu = info.agentOwner;
    std::pair<std::vector<hsa_agent_t>, std::vector<hsa_agent_t>> agents;
    if (hsa_iterate_agents(collect_agents, &agents) != HSA_STATUS_SUCCESS || agents.second.empty()) return nullptr;
    bool is_gpu = false;
    for (hsa_agent_t a : agents.first) is_gpu |= (a.handle == e->gpu.handle);
    if (!is_gpu) return nullptr;
    e->cpu = agents.second[0];
    hsa_amd_pointer_info_t host_info{}; host_info.size = sizeof(host_info);
    if (hsa_amd_pointer_info(host, &host_info, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS)
        for (hsa_agent_t a : agents.second) if (a.handle == host_info.agentOwner.handle) e->cpu = a;      // the socket the ring lives on
    uint32_t free_mask = 0, preferred = 0;
    // (informative only: a busy engine still takes the copy into its queue — the call must work, the mask need not be non-zero)
    if (hsa_amd_memory_copy_engine_status(e->cpu, e->gpu, &free_mask) != HSA_STATUS_SUCCESS && hsa_amd_memory_get_preferred_copy_engine(e->cpu, e->gpu, &preferred) != HSA_STATUS_SUCCESS) return nullptr;
    if (hsa_amd_memory_get_preferred_copy_engine(e->cpu, e->gpu, &preferred) != HSA_STATUS_SUCCESS) preferred = 0;
    uint32_t pick = __builtin_popcount(preferred) >= 2 ? preferred : 0x3u;
    e->engine[0] = pick & (~pick + 1u);                              // lowest set bit
    const uint32_t rest = pick & (pick - 1u);
    e->engine[1] = rest ? (rest & (~rest + 1u)) : e->engine[0];
    int a = -1, b = -1;
    if (const char* named = getenv("SHADERFLOW_SDMA_ENGINES")) if (sscanf(named, "%d,%d", &a, &b) == 2 && a >= 0 && a < 16 && b >= 0 && b < 16) { e->engine[0] = 1u << a; e->engine[1] = 1u << b; }
    e->usable = true;
    return e;
}

// The same for a PEER copy (sharded export, "device-sdma"): the destination is a window another process exported (sfx_peer_open) — its
// owner is another GPU of the node (or, in the one-GPU tests, this one). The engines are the ones HSA recommends for that ordered
// pair of agents: on the node's fully connected fabric every peer has its own xGMI link and the runtime pairs links with SDMA
// engines, so NAMING them keeps two copies of one rank on the engines of ITS link instead of on whichever engine is idle
// (hipMemcpyAsync's lottery). SHADERFLOW_PEER_ENGINES=a,b overrides, SHADERFLOW_PEER=hip keeps HIP's copy streams.
static EngineCopy* peer_engine_copy(Context* c, const void* remote, const void* local) {
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    if (c->peer_engines) return c->peer_engines->usable ? c->peer_engines : nullptr;
    EngineCopy* e = c->peer_engines = new EngineCopy();
    // OPT-IN (SHADERFLOW_PEER=engine) since round 6: the named-engine route has only ever copied into its own process' window on ONE
    // GPU. Until it has run between two real GPUs the default is HIP's copy streams (hipMemcpyAsync on the context's probed copy streams),
    // which every ROCm release exercises; bench.py's "sdma" legs ask for the engines explicitly after their collective preflight.
    const char* route = getenv("SHADERFLOW_PEER");
    if (!route || strcmp(route, "engine")) return nullptr;
    if (hsa_init() != HSA_STATUS_SUCCESS) return nullptr;
    hsa_amd_pointer_info_t here{}, there{};
    here.size = sizeof(here); there.size = sizeof(there);
    if (hsa_amd_pointer_info(local, &here, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || here.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return nullptr;
    if (hsa_amd_pointer_info(remote, &there, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || there.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return nullptr;
    e->gpu = here.agentOwner; e->cpu = there.agentOwner;            // (`cpu` = the destination's agent: the GPU that owns the window)
    std::pair<std::vector<hsa_agent_t>, std::vector<hsa_agent_t>> agents;
    if (hsa_iterate_agents(collect_agents, &agents) != HSA_STATUS_SUCCESS) return nullptr;
    bool source_known = false, target_known = false;
    for (hsa_agent_t a : agents.first) { source_known |= (a.handle == e->gpu.handle); target_known |= (a.handle == e->cpu.handle); }
    if (!source_known || !target_known) return nullptr;
    uint32_t preferred = 0, free_mask = 0;
    if (hsa_amd_memory_get_preferred_copy_engine(e->cpu, e->gpu, &preferred) != HSA_STATUS_SUCCESS) preferred = 0;
    if (!preferred && hsa_amd_memory_copy_engine_status(e->cpu, e->gpu, &free_mask) == HSA_STATUS_SUCCESS) preferred = free_mask;
    if (!preferred) return nullptr;                                   // no engine serves the pair (HSA's own choice would be a blit kernel)
    e->engine[0] = preferred & (~preferred + 1u);
    const uint32_t rest = preferred & (preferred - 1u);
    e->engine[1] = rest ? (rest & (~rest + 1u)) : e->engine[0];
    int a = -1, b = -1;
    if (const char* named = getenv("SHADERFLOW_PEER_ENGINES")) if (sscanf(named, "%d,%d", &a, &b) == 2 && a >= 0 && a < 16 && b >= 0 && b < 16) { e->engine[0] = 1u << a; e->engine[1] = 1u << b; }
    e->usable = true;
    return e;
}

// device-to-device copy of a frame as a KERNEL on the caller's stream (hipMemcpyAsync hands it to a copy engine: measured 5.6 GB/s for
// a 6.2 MB frame inside the frame loop — 1.1 ms per frame; this is 2-3 us)
typedef unsigned int frame_u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_frame_copy(const frame_u4* __restrict__ src, frame_u4* __restrict__ dst, size_t n16, const unsigned char* __restrict__ src_tail,
                                                    unsigned char* __restrict__ dst_tail, int tail) {
    for (size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x*blockDim.x) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
static hipError_t frame_copy(void* dst, const void* src, size_t nbytes, hipStream_t stream) {
    if (((uintptr_t)dst | (uintptr_t)src) & 15) return hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, stream);
    const size_t n16 = nbytes/16;
    const unsigned blocks = (unsigned)std::min<size_t>(2048, (n16 + 255)/256 + 1);
    hipLaunchKernelGGL(k_frame_copy, dim3(blocks), dim3(256), 0, stream, (const frame_u4*)src, (frame_u4*)dst, n16, (const unsigned char*)src + n16*16,
                       (unsigned char*)dst + n16*16, (int)(nbytes - n16*16));
    return hipGetLastError();
}

// A frame is complete on the render stream: polled for ~200 us before the thread blocks (hipEventSynchronize wakes up late: see finish())
// false: the event reported an error (a sticky one of the runtime, a failed launch before it): the frame must NOT be treated as rendered
static bool wait_frame_ready(hipEvent_t event) {
    const auto started = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t state = hipEventQuery(event);
        if (state == hipSuccess) return true;
        if (state != hipErrorNotReady) { (void)hipGetLastError(); return false; }
        if (std::chrono::steady_clock::now() - started > std::chrono::microseconds(200)) {
            if (hipEventSynchronize(event) == hipSuccess) return true;
            (void)hipGetLastError();
            return false;
        }
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
}

struct EngineLanes {
    Context* c = nullptr;
    EngineCopy* e = nullptr;                                        // null: the lanes are the context's copy streams
    bool resolved = false;
    // up to four copies in flight: lanes 0 and 2 on one engine, 1 and 3 on the other. An engine takes its second copy from its own
    // queue the moment the first ends; with one copy per engine both ended together and the link idled until the host had issued the
    // next pair (1080p frames: 137 us per frame where the link needs 112)
    static constexpr int LANES = 4;
    hsa_signal_t done[LANES] = {};
    bool busy[LANES] = {};
    bool via_hsa[LANES] = {};                                       // the route the lane's copy in flight was issued on: it is finished on that one
    hipMemcpyKind kind = hipMemcpyDeviceToHost;                     // the HIP route's direction (read-out, or a peer copy)

    // `host` / `device`: the first frame's two ends (they name the agents). Called from the thread that issues.
    void resolve(Context* context, const void* host, const void* device, bool peer = false) {
        if (resolved) return;
        resolved = true; c = context;
        e = peer ? peer_engine_copy(context, host, device) : engine_copy(context, host, device);
        kind = peer ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        if (e) for (auto& signal : done) if (hsa_signal_create(0, 0, nullptr, &signal) != HSA_STATUS_SUCCESS) { e = nullptr; break; }
        if (!e) (void)context_copy_streams(context);                  // the lanes are HIP copy streams then
    }
    // false: the copy could not be queued on any route
    bool issue(int lane, void* host, const void* device, size_t nbytes) {
        if (e && !done[lane].handle) e = nullptr;                      // (a retired signal could not be replaced: HIP's copies from here on)
        if (e) {
            hsa_signal_store_relaxed(done[lane], 1);
            hsa_status_t status = hsa_amd_memory_async_copy_on_engine(host, e->cpu, device, e->gpu, nbytes, 0, nullptr, done[lane], (hsa_amd_sdma_engine_id_t)e->engine[lane & 1], false);
            if (status != HSA_STATUS_SUCCESS) {                      // the engine's queue could not be had: let HSA choose
                hsa_signal_store_relaxed(done[lane], 1);
                status = hsa_amd_memory_async_copy(host, e->cpu, device, e->gpu, nbytes, 0, nullptr, done[lane]);
            }
            if (status == HSA_STATUS_SUCCESS) { busy[lane] = true; via_hsa[lane] = true; return true; }
            // HSA refuses: HIP's copies from here on. The lanes still in flight through HSA keep their route (via_hsa) and are drained
            // on it by finish(); nothing of theirs ever ran on the copy streams.
            e = nullptr;
            if (context_copy_streams(c) != SFX_OK) return false;
        }
        if (!c->copy_streams[0] && context_copy_streams(c) != SFX_OK) return false;
        if (hipMemcpyAsync(host, device, nbytes, kind, c->copy_streams[lane & 1]) != hipSuccess) { (void)hipGetLastError(); return false; }
        busy[lane] = true; via_hsa[lane] = false;
        return true;
    }
    // ~200 us in HSA's timestamp ticks (the hint of an ACTIVE wait)
    static uint64_t poll_ticks() {
        static const uint64_t ticks = [] { uint64_t hz = 0; return (hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &hz) == HSA_STATUS_SUCCESS && hz) ? hz/5000 : 20000; }();
        return ticks;
    }
    // has the lane's copy ended (either way)? never blocks
    bool landed(int lane) const {
        if (!busy[lane]) return true;
        if (via_hsa[lane]) return hsa_signal_load_scacquire(done[lane]) < 1;
        return hipStreamQuery(c->copy_streams[lane & 1]) != hipErrorNotReady;
    }
    // false: the copy FAILED (the runtime left its signal negative, or the stream reports an error): the slot's bytes are not the frame
    bool finish(int lane) {
        if (!busy[lane]) return true;
        bool ok = true;
        if (via_hsa[lane]) {
            // the wait may return before the condition holds (the specification allows spurious returns): ask again until it does;
            // a failed copy leaves the signal NEGATIVE, which satisfies "< 1" as well, so the value itself is looked at.
            // Polled first: a blocked wait is woken by an interrupt tens of microseconds after the copy ended, which at 1080p (a frame
            // every 60-110 us) is a large part of the frame; after ~200 us of polling the thread blocks like before.
            // The blocked wait is bounded (SHADERFLOW_COPY_TIMEOUT seconds, default 120): a copy whose signal never moves — an engine
            // that does not reach the other agent: the peer copies have only ever run on one GPU — is REPORTED as failed instead of
            // hanging its export. (Its signal is then left alone: the engine may still write it.)
            if (hsa_signal_wait_scacquire(done[lane], HSA_SIGNAL_CONDITION_LT, 1, poll_ticks(), HSA_WAIT_STATE_ACTIVE) >= 1) {
                static const int limit = [] { const char* e = getenv("SHADERFLOW_COPY_TIMEOUT"); const int v = e ? atoi(e) : 120; return v > 0 ? v : 120; }();
                int seconds = 0;
                while (hsa_signal_wait_scacquire(done[lane], HSA_SIGNAL_CONDITION_LT, 1, poll_ticks()*5000, HSA_WAIT_STATE_BLOCKED) >= 1)
                    if (++seconds >= limit) break;
            }
            const hsa_signal_value_t left = hsa_signal_load_relaxed(done[lane]);
            ok = left == 0;
            if (left >= 1) {
                // TIMED OUT: the engine may still decrement this signal whenever its copy ends — armed again for the lane's next copy it
                // would make that copy look complete before it is. The signal is RETIRED (left to the late copy, never destroyed or reused)
                // and the lane gets a fresh one; when none can be had the lane leaves the engine route.
                hsa_signal_t fresh{};
                if (hsa_signal_create(0, 0, nullptr, &fresh) == HSA_STATUS_SUCCESS) done[lane] = fresh;
                else { done[lane] = hsa_signal_t{}; e = nullptr; }
            }
        } else {
            ok = hipStreamSynchronize(c->copy_streams[lane & 1]) == hipSuccess;   // (a stream's later copy too: in order, so nothing is released early)
            if (!ok) (void)hipGetLastError();
        }
        busy[lane] = false;
        return ok;
    }
    void release() {
        for (int lane = 0; lane < LANES; lane++) finish(lane);
        if (resolved && done[0].handle) for (auto& signal : done) if (signal.handle) hsa_signal_destroy(signal);
        for (auto& signal : done) signal = hsa_signal_t{};
    }
};

// how the copy streams of this context were chosen: streams looked at, and how many of them ran in series with the render stream
extern "C" int sfx_ctx_copy_streams(sfx_handle h, int* candidates, int* colliding) {
    CTX_OR_FAIL(c, h);
    USE_DEVICE(c);
    int rc = context_copy_streams(c);
    if (rc) return rc;
    if (candidates) *candidates = c->copy_candidates;
    if (colliding) *colliding = c->copy_colliding;
    return SFX_OK;
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
    for (hipStream_t stream : c->copy_streams) if (stream) { hipStreamSynchronize(stream); hipStreamDestroy(stream); }
    delete c->engines;
    peer_stop(c);
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
extern "C" int sfx_device_copy(sfx_handle h, void* dst, const void* src, size_t nbytes) {
    CTX_OR_FAIL(c, h);
    if (!dst || !src) return fail(SFX_E_INVALID, "null device pointer");
    USE_DEVICE(c);
    HIP_TRY(frame_copy(dst, src, nbytes, c->stream));              // a kernel on the context's stream, not a copy engine
    return SFX_OK;
}

// ---- peer windows: frames from this rank's HBM straight into another process' buffer, on the copy engines ----------------------
// The sharded export's gather without a collective and without compute units (DESIGN.md §6 "device-sdma"): rank 0 exports its
// resident frame buffer as an IPC handle, every other rank maps it and copies its finished frames to where they belong with
// hipMemcpyAsync on a copy stream — SDMA engines over the rank's own xGMI link, concurrent with the next batch's kernels.
extern "C" int sfx_peer_export(sfx_handle h, void* device_ptr, void* handle64) {
    CTX_OR_FAIL(c, h);
    if (!device_ptr || !handle64) return fail(SFX_E_INVALID, "peer window: null pointer");
    USE_DEVICE(c);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI carries IPC handles as 64 opaque bytes");
    HIP_TRY(hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, device_ptr));
    return SFX_OK;
}
extern "C" int sfx_peer_open(sfx_handle h, const void* handle64, void** device_ptr) {
    CTX_OR_FAIL(c, h);
    if (!device_ptr || !handle64) return fail(SFX_E_INVALID, "peer window: null pointer");
    USE_DEVICE(c);
    hipIpcMemHandle_t handle;
    memcpy(&handle, handle64, sizeof handle);
    HIP_TRY(hipIpcOpenMemHandle(device_ptr, handle, hipIpcMemLazyEnablePeerAccess));
    return SFX_OK;
}
// Peer copies are issued by a thread of the context, like the read-out's (EngineLanes): it waits ON THE HOST for the source to be
// complete on the render stream (an event per tag), hands the copy to one of the two SDMA engines HSA recommends for the pair of
// GPUs — up to four in flight, two per engine — and waits on HSA's signals. Nothing passes through a HIP queue, so no hardware-queue
// sharing with the render stream and no engine lottery; where HSA does not answer the lanes are HIP's copy streams.
struct PeerCopier {
    Context* ctx = nullptr;
    struct Job { void* dst; const void* src; size_t nbytes; int tag; };
    std::thread worker;
    std::mutex mutex;
    std::condition_variable wake, idle;
    std::deque<Job> queue;
    EngineLanes lanes;
    hipEvent_t ready[16] = {};
    int pending[16] = {};                                            // copies of a tag queued or in flight
    int error = 0;
    bool stop = false;
    uint64_t copies = 0, bytes = 0;
};

static void peer_copier(PeerCopier* p) {
    hipSetDevice(p->ctx->device);
    int in_lane[EngineLanes::LANES] = {-1, -1, -1, -1}, next = 0;
    auto finish = [&](int lane) {
        if (in_lane[lane] < 0) return;
        const bool ok = p->lanes.finish(lane);
        { std::lock_guard<std::mutex> lock(p->mutex); if (!ok) p->error = 1; p->pending[in_lane[lane]]--; }
        in_lane[lane] = -1;
        p->idle.notify_all();
    };
    for (;;) {
        PeerCopier::Job job;
        {
            std::unique_lock<std::mutex> lock(p->mutex);
            if (p->queue.empty()) {
                lock.unlock();
                for (int k = 0; k < EngineLanes::LANES; k++) finish((next + k) % EngineLanes::LANES);
                lock.lock();
                p->wake.wait(lock, [&] { return p->stop || !p->queue.empty(); });
                if (p->queue.empty()) return;
            }
            job = p->queue.front(); p->queue.pop_front();
        }
        for (int lane = 0; lane < EngineLanes::LANES; lane++) if (in_lane[lane] >= 0 && p->lanes.landed(lane)) finish(lane);
        const bool rendered = wait_frame_ready(p->ready[job.tag]);   // the source is complete on the render stream (false: its event reports an error — nothing is copied)
        finish(next);
        p->lanes.resolve(p->ctx, job.dst, job.src, true);
        if (!rendered || !p->lanes.issue(next, job.dst, job.src, job.nbytes)) {
            std::lock_guard<std::mutex> lock(p->mutex);
            p->error = 1; p->pending[job.tag]--;
            p->idle.notify_all();
            continue;
        }
        in_lane[next] = job.tag;
        next = (next + 1) % EngineLanes::LANES;
    }
}

static void peer_stop(Context* c) {
    PeerCopier* p = c->peer;
    if (!p) return;
    { std::lock_guard<std::mutex> lock(p->mutex); p->stop = true; }
    p->wake.notify_all();
    if (p->worker.joinable()) p->worker.join();
    p->lanes.release();
    for (auto& e : p->ready) if (e) hipEventDestroy(e);
    { auto& list = c->readouts; list.erase(std::remove_if(list.begin(), list.end(), [&](const std::pair<void*, void (*)(void*)>& e) { return e.first == p; }), list.end()); }
    delete p;
    c->peer = nullptr;
}
static int peer_wait(PeerCopier* p, int tag) {                      // tag < 0: every tag
    std::unique_lock<std::mutex> lock(p->mutex);
    p->idle.wait(lock, [&] { if (tag >= 0) return p->pending[tag] == 0; for (int n : p->pending) if (n) return false; return true; });
    return p->error ? fail(SFX_E_HIP, "peer copy: neither HSA nor HIP completed the copy") : SFX_OK;
}

extern "C" int sfx_peer_close(sfx_handle h, void* device_ptr) {
    CTX_OR_FAIL(c, h);
    USE_DEVICE(c);
    if (c->peer) { const int rc = peer_wait(c->peer, -1); if (rc) return rc; }
    HIP_TRY(hipIpcCloseMemHandle(device_ptr));
    return SFX_OK;
}
// `nbytes` from `local_src` (complete on the context's stream once everything queued there so far has run) to `remote_dst` (inside a
// window opened with sfx_peer_open, or any device pointer): asynchronous. `lane` (0..15) tags the source buffer for sfx_peer_fence.
extern "C" int sfx_peer_copy(sfx_handle h, void* remote_dst, const void* local_src, size_t nbytes, int lane) {
    CTX_OR_FAIL(c, h);
    if (!remote_dst || !local_src || lane < 0 || lane >= 16) return fail(SFX_E_INVALID, "peer copy: pointers / lane %d", lane);
    USE_DEVICE(c);
    if (!c->peer) {
        c->peer = new PeerCopier();
        c->peer->ctx = c;
        c->peer->worker = std::thread(peer_copier, c->peer);
        c->readouts.push_back({c->peer, [](void* copier) { (void)peer_wait((PeerCopier*)copier, -1); }});   // sfx_device_free waits for copies in flight
    }
    PeerCopier* p = c->peer;
    { const int rc = peer_wait(p, lane); if (rc) return rc; }      // the tag's event is recorded again below: its previous copy must have taken it
    if (!p->ready[lane]) HIP_TRY(hipEventCreateWithFlags(&p->ready[lane], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(p->ready[lane], c->stream));
    {
        std::lock_guard<std::mutex> lock(p->mutex);
        p->pending[lane]++;
        p->queue.push_back({remote_dst, local_src, nbytes, lane});
        p->copies++; p->bytes += nbytes;
    }
    p->wake.notify_one();
    return SFX_OK;
}
// the last copy tagged `lane` has left its source (which may then be overwritten). A HOST wait since round 5 — the copies run outside
// HIP's queues, there is no event a stream could wait for; a pipelined sender asks about a copy it queued a whole step ago.
extern "C" int sfx_peer_fence(sfx_handle h, int lane) {
    CTX_OR_FAIL(c, h);
    if (lane < 0 || lane >= 16) return fail(SFX_E_INVALID, "peer fence: lane %d", lane);
    return c->peer ? peer_wait(c->peer, lane) : SFX_OK;
}
// every copy issued so far has landed (host wait)
extern "C" int sfx_peer_flush(sfx_handle h) {
    CTX_OR_FAIL(c, h);
    return c->peer ? peer_wait(c->peer, -1) : SFX_OK;
}
// How this context's peer copies travel: *via_engines 1 = SDMA engines named through HSA (engine_ids: their indices), 0 = HIP's copy
// streams (HSA did not answer, or SHADERFLOW_PEER=hip), -1 = no copy has been issued yet. For measurements and their records.
extern "C" int sfx_peer_route(sfx_handle h, int* via_engines, int* engine_ids /* [2] */, unsigned long long* copies, unsigned long long* bytes) {
    CTX_OR_FAIL(c, h);
    PeerCopier* p = c->peer;
    const bool resolved = p && p->lanes.resolved;
    if (via_engines) *via_engines = !resolved ? -1 : (p->lanes.e ? 1 : 0);
    if (engine_ids) for (int k = 0; k < 2; k++) engine_ids[k] = (resolved && p->lanes.e) ? __builtin_ctz(p->lanes.e->engine[k] ? p->lanes.e->engine[k] : 1u) : -1;
    if (copies) *copies = p ? p->copies : 0;
    if (bytes) *bytes = p ? p->bytes : 0;
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

static int launch_status() {
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
// Read-out ring with a pipe writer thread (turbopipe's role, exporting.py:147-171)

struct Ring : Object {
    Context* ctx;
    size_t frame_bytes;
    int slots;
    std::vector<void*> host;
    std::vector<void*> staging;                                     // device copies of texture reads (sfx_ring_read_async), allocated on first use
    std::vector<hipEvent_t> ready;                                  // per slot: recorded on the render stream when the slot's frame is complete
    hipEvent_t fences[2];
    // copier: waits for a frame on the host, copies it on an engine lane (two in flight), marks the slot copied
    struct CopyJob { int slot; const void* source; hipEvent_t ready; };
    std::thread copier;
    std::deque<CopyJob> copy_queue;
    std::vector<int> copying;                       // 1 while the slot's copy is queued or in flight
    EngineLanes lanes;
    int lane_count = 2;
    int copy_error = 0;
    // writer
    std::thread writer;
    std::mutex mutex;
    std::condition_variable wake, idle, copy_wake;
    std::deque<std::pair<int, int>> queue;          // (slot, fd)
    std::vector<int> pending;                       // writes queued or running per slot
    bool stop = false;
    int io_error = 0;
};

static void ring_copier(Ring* r) {
    hipSetDevice(r->ctx->device);
    int in_lane[EngineLanes::LANES] = {-1, -1, -1, -1}, next = 0;
    auto finish = [&](int lane) {
        if (in_lane[lane] < 0) return;
        const bool ok = r->lanes.finish(lane);
        { std::lock_guard<std::mutex> lock(r->mutex); if (!ok) r->copy_error = 1; r->copying[in_lane[lane]] = 0; }
        in_lane[lane] = -1;
        r->idle.notify_all();
    };
    // frames that have landed are handed to the writer at once, not when their lane comes round again (a continuously fed queue
    // never runs empty, and the copier is about to block on the NEXT frame's render)
    auto release_landed = [&] { for (int lane = 0; lane < r->lane_count; lane++) if (in_lane[lane] >= 0 && r->lanes.landed(lane)) finish(lane); };
    for (;;) {
        Ring::CopyJob job;
        {
            std::unique_lock<std::mutex> lock(r->mutex);
            if (r->copy_queue.empty()) {                             // nothing to issue: let what is in flight land (in issue order), then sleep
                lock.unlock();
                for (int k = 0; k < r->lane_count; k++) finish((next + k) % r->lane_count);
                lock.lock();
                r->copy_wake.wait(lock, [&] { return r->stop || !r->copy_queue.empty(); });
                if (r->copy_queue.empty()) return;
            }
            job = r->copy_queue.front(); r->copy_queue.pop_front();
        }
        static const bool trace = getenv("SHADERFLOW_RING_TRACE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        release_landed();
        const bool rendered = wait_frame_ready(job.ready);           // the frame is complete on the render stream (false: its event reports an error — the slot is failed, not filled)
        const auto t1 = std::chrono::steady_clock::now();
        release_landed();
        finish(next);                                                // the lane's previous copy
        const auto t2 = std::chrono::steady_clock::now();
        if (trace) fprintf(stderr, "ring copier: slot %d event wait %.0f us, lane finish %.0f us\n", job.slot, std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count());
        r->lanes.resolve(r->ctx, r->host[job.slot], job.source);
        if (!rendered || !r->lanes.issue(next, r->host[job.slot], job.source, r->frame_bytes)) {
            std::lock_guard<std::mutex> lock(r->mutex);
            r->copy_error = 1; r->copying[job.slot] = 0;
            r->idle.notify_all();
            continue;
        }
        in_lane[next] = job.slot;
        next = (next + 1) % r->lane_count;
    }
}

static void ring_writer(Ring* r) {
    for (;;) {
        std::pair<int, int> job;
        {
            std::unique_lock<std::mutex> lock(r->mutex);
            r->wake.wait(lock, [&] { return r->stop || !r->queue.empty(); });
            if (r->queue.empty()) return;
            job = r->queue.front(); r->queue.pop_front();
            r->idle.wait(lock, [&] { return r->copying[job.first] == 0; });      // the frame has landed in the slot's host buffer
        }
        const char* p = (const char*)r->host[job.first];
        size_t left = r->frame_bytes;
        int err = 0;
        while (left > 0) {
            ssize_t n = ::write(job.second, p, left);
            if (n < 0) { if (errno == EINTR) continue; err = errno; break; }
            p += n; left -= (size_t)n;
        }
        {
            std::lock_guard<std::mutex> lock(r->mutex);
            if (err) r->io_error = err;
            r->pending[job.first]--;
        }
        r->idle.notify_all();
    }
}

extern "C" int sfx_ring_create(sfx_handle h, size_t frame_bytes, int slots, sfx_handle* out) {
    CTX_OR_FAIL(c, h);
    if (!out || slots < 1 || slots > 64 || frame_bytes == 0) return fail(SFX_E_INVALID, "ring of %d slots x %zu bytes", slots, frame_bytes);
    USE_DEVICE(c);
    Ring* r = new Ring();
    r->magic = MAGIC_RING; r->ctx = c; r->frame_bytes = frame_bytes; r->slots = slots;
    r->host.resize(slots); r->ready.resize(slots); r->pending.assign(slots, 0); r->copying.assign(slots, 0);
    r->lane_count = std::min(EngineLanes::LANES, std::max(1, slots - 1));                             // (a slot is being filled or written while the others land)
    if (const char* n = getenv("SHADERFLOW_COPY_STREAMS")) r->lane_count = std::min(r->lane_count, std::max(1, atoi(n)));   // A/B switch for measurements
    for (auto& f : r->fences) HIP_TRY(hipEventCreateWithFlags(&f, hipEventDisableTiming));
    for (int k = 0; k < slots; k++) {
        HIP_TRY(hipHostMalloc(&r->host[k], frame_bytes, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&r->ready[k], hipEventDisableTiming));
    }
    r->copier = std::thread(ring_copier, r);
    r->writer = std::thread(ring_writer, r);
    c->readouts.push_back({r, [](void* ring) {
        Ring* q = (Ring*)ring;
        std::unique_lock<std::mutex> lock(q->mutex);
        q->idle.wait(lock, [&] { for (int busy : q->copying) if (busy) return false; return true; });
    }});
    *out = handle_of(r);
    return SFX_OK;
}

// the slot's last frame has been written out (turbopipe.sync(buffer) before reuse) AND any copy into it has landed
static int ring_wait_slot(Ring* r, int slot) {
    std::unique_lock<std::mutex> lock(r->mutex);
    r->idle.wait(lock, [&] { return r->pending[slot] == 0 && r->copying[slot] == 0; });
    if (r->copy_error) return fail(SFX_E_HIP, "frame read-out: neither HSA nor HIP accepted the copy");
    return r->io_error ? fail(SFX_E_IO, "pipe write failed: %s", strerror(r->io_error)) : SFX_OK;
}
static int ring_wait_copy(Ring* r, int slot) {
    std::unique_lock<std::mutex> lock(r->mutex);
    r->idle.wait(lock, [&] { return r->copying[slot] == 0; });
    return r->copy_error ? fail(SFX_E_HIP, "frame read-out: neither HSA nor HIP accepted the copy") : SFX_OK;
}
static int ring_queue_copy(Ring* r, const void* dptr, int slot, hipEvent_t ready) {
    {
        std::lock_guard<std::mutex> lock(r->mutex);
        r->copying[slot] = 1;
        r->copy_queue.push_back({slot, dptr, ready});
    }
    r->copy_wake.notify_one();
    return SFX_OK;
}

extern "C" int sfx_ring_read_device_async(sfx_handle h, const void* dptr, int slot) {
    Ring* r = get<Ring>(h, MAGIC_RING);
    if (!r || !dptr || slot < 0 || slot >= r->slots) return fail(SFX_E_INVALID, "invalid ring handle, pointer or slot");
    USE_DEVICE(r->ctx);
    int rc = ring_wait_slot(r, slot);                               // turbopipe.sync(buffer) before reuse
    if (rc) return rc;
    HIP_TRY(hipEventRecord(r->ready[slot], r->ctx->stream));        // the frame is complete on the render stream when this event is
    return ring_queue_copy(r, dptr, slot, r->ready[slot]);
}

extern "C" int sfx_ring_fence(sfx_handle h, int which) {
    Ring* r = get<Ring>(h, MAGIC_RING);
    if (!r || which < 0 || which > 1) return fail(SFX_E_INVALID, "invalid ring handle or fence");
    USE_DEVICE(r->ctx);
    HIP_TRY(hipEventRecord(r->fences[which], r->ctx->stream));
    return SFX_OK;
}

extern "C" int sfx_ring_read_fenced_async(sfx_handle h, const void* dptr, int slot, int which) {
    Ring* r = get<Ring>(h, MAGIC_RING);
    if (!r || !dptr || slot < 0 || slot >= r->slots || which < 0 || which > 1) return fail(SFX_E_INVALID, "invalid ring handle, pointer, slot or fence");
    USE_DEVICE(r->ctx);
    int rc = ring_wait_slot(r, slot);
    if (rc) return rc;
    return ring_queue_copy(r, dptr, slot, r->fences[which]);
}

// The frame read into `slot` has left its device buffer: the caller may render into that buffer again when this returns. (Rounds 1-3
// made the render STREAM wait for a copy event; the read-out no longer runs on a HIP stream, so the host waits — by the time a
// pipelined export asks, the copy it names finished a whole batch ago.)
extern "C" int sfx_ring_stream_wait(sfx_handle h, int slot) {
    Ring* r = get<Ring>(h, MAGIC_RING);
    if (!r || slot < 0 || slot >= r->slots) return fail(SFX_E_INVALID, "invalid ring handle or slot");
    return ring_wait_copy(r, slot);
}

// fbo.read_into(buffer) is a GL command: it has read the texture before the next draw call touches it. Here the read-out runs beside
// the render stream, so the frame is first copied — on the render stream, in order with the draws — into a device buffer of the slot,
// and the asynchronous read-out takes it from there (the frame loop renders the next frame into the same texture right away).
extern "C" int sfx_ring_read_async(sfx_handle h, sfx_handle tex, int slot) {
    Ring* r = get<Ring>(h, MAGIC_RING);
    Texture* t = get<Texture>(tex, MAGIC_TEX);
    if (!r || !t || slot < 0 || slot >= r->slots) return fail(SFX_E_INVALID, "invalid ring or texture handle, or slot");
    if (t->nbytes != r->frame_bytes) return fail(SFX_E_INVALID, "texture holds %zu bytes, ring slots %zu", t->nbytes, r->frame_bytes);
    USE_DEVICE(r->ctx);
    int rc = ring_wait_slot(r, slot);                               // the slot's last frame has left its staging buffer too
    if (rc) return rc;
    if (r->staging.empty()) r->staging.assign(r->slots, nullptr);
    if (!r->staging[slot]) HIP_TRY(hipMalloc(&r->staging[slot], r->frame_bytes));
    HIP_TRY(frame_copy(r->staging[slot], t->data, r->frame_bytes, r->ctx->stream));
    HIP_TRY(hipEventRecord(r->ready[slot], r->ctx->stream));
    return ring_queue_copy(r, r->staging[slot], slot, r->ready[slot]);
}

extern "C" int sfx_ring_sync(sfx_handle h, int slot, void** host_ptr) {
    Ring* r = get<Ring>(h, MAGIC_RING);
    if (!r || slot < 0 || slot >= r->slots) return fail(SFX_E_INVALID, "invalid ring handle or slot");
    int rc = ring_wait_copy(r, slot);
    if (rc) return rc;
    if (host_ptr) *host_ptr = r->host[slot];
    return SFX_OK;
}

extern "C" int sfx_ring_pipe(sfx_handle h, int slot, int fd) {
    Ring* r = get<Ring>(h, MAGIC_RING);
    if (!r || slot < 0 || slot >= r->slots || fd < 0) return fail(SFX_E_INVALID, "invalid ring handle, slot or fd");
    {
        std::lock_guard<std::mutex> lock(r->mutex);
        if (r->io_error) return fail(SFX_E_IO, "pipe write failed: %s", strerror(r->io_error));
        r->pending[slot]++;
        r->queue.push_back({slot, fd});
    }
    r->wake.notify_one();
    return SFX_OK;
}

// `count` frames of a batch, `stride` bytes apart, through consecutive slots from `first_slot`: read (after fence `which`, or after
// everything queued on the render stream so far when which < 0) and piped to `fd` — one native call instead of 2·count
extern "C" int sfx_ring_pipe_frames(sfx_handle h, const void* dptr, size_t stride, int count, int first_slot, int which, int fd) {
    Ring* r = get<Ring>(h, MAGIC_RING);
    if (!r || !dptr || count < 0 || first_slot < 0 || which > 1 || fd < 0) return fail(SFX_E_INVALID, "invalid ring handle, pointer, count, slot, fence or fd");
    for (int k = 0; k < count; k++) {
        const int slot = (first_slot + k) % r->slots;
        const void* frame = (const char*)dptr + (size_t)k*stride;
        int rc = which < 0 ? sfx_ring_read_device_async(h, frame, slot) : sfx_ring_read_fenced_async(h, frame, slot, which);
        if (!rc) rc = sfx_ring_pipe(h, slot, fd);
        if (rc) return rc;
    }
    return SFX_OK;
}

extern "C" int sfx_ring_pipe_sync(sfx_handle h, int slot) {
    Ring* r = get<Ring>(h, MAGIC_RING);
    if (!r || slot >= r->slots) return fail(SFX_E_INVALID, "invalid ring handle or slot");
    if (slot >= 0) return ring_wait_slot(r, slot);
    for (int k = 0; k < r->slots; k++) { int rc = ring_wait_slot(r, k); if (rc) return rc; }
    return SFX_OK;
}

extern "C" int sfx_ring_destroy(sfx_handle h) {
    Ring* r = get<Ring>(h, MAGIC_RING);
    if (!r) return fail(SFX_E_INVALID, "invalid ring handle");
    { std::lock_guard<std::mutex> lock(r->mutex); r->stop = true; }
    r->wake.notify_all(); r->copy_wake.notify_all();
    if (r->copier.joinable()) r->copier.join();
    if (r->writer.joinable()) r->writer.join();
    hipSetDevice(r->ctx->device);
    r->lanes.release();
    { auto& list = r->ctx->readouts; list.erase(std::remove_if(list.begin(), list.end(), [&](const std::pair<void*, void (*)(void*)>& e) { return e.first == r; }), list.end()); }
    for (int k = 0; k < r->slots; k++) { hipHostFree(r->host[k]); hipEventDestroy(r->ready[k]); }
    for (void* p : r->staging) if (p) hipFree(p);
    for (auto& f : r->fences) hipEventDestroy(f);
    r->magic = 0;
    delete r;
    return SFX_OK;
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
    Ring* ring = hring ? get<Ring>(hring, MAGIC_RING) : nullptr;
    if (!passes || npasses < 1 || nmatrices < 0 || (nmatrices && !matrices) || !clock || nframes < 0) return fail(SFX_E_INVALID, "clock sequence: null tables");
    if (hring && !ring) return fail(SFX_E_INVALID, "clock sequence: invalid ring handle");
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
        if (ring && fd >= 0) {                                      // exporting.pipe (exporting.py:151-174)
            const int slot = (first_slot + f) % ring->slots;
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

// ---------------------------------------------------------------------------------------------------------
// Encoder hand-off, optional half (exporting.py:94-134 hands rgb24 to ffmpeg, whose swscale converts to the codec's
// yuv420p on the CPU): planar 4:2:0 on the device halves what crosses PCIe and the pipe. The arithmetic is DEFINED here (no ffmpeg
// binary exists in this environment to pin swscale's against): BT.601 limited range in the classic 8-bit integer form,
//   Y = ((66 R + 129 G + 25 B + 128) >> 8) + 16 per pixel; chroma from the rounded mean of the 2 x 2 block's R, G, B
//   ((sum + 2) >> 2): U = ((-38 R - 74 G + 112 B + 128) >> 8) + 128, V = ((112 R - 94 G - 18 B + 128) >> 8) + 128
// (arithmetic shifts; `matrix` 1: BT.709 limited, 47/157/16, -26/-86/112, 112/-102/-10). Rows keep the RGB frame's order.
// Layout I420: Y (h rows of w), U (h/2 rows of w/2), V. The parity oracle restates it in C (sfo_rgb_to_yuv420).
struct YuvMatrix { int yr, yg, yb, ur, ug, ub, vr, vg, vb; };
__device__ __forceinline__ YuvMatrix yuv_matrix(int matrix) {
    return matrix == 1 ? YuvMatrix{47, 157, 16, -26, -86, 112, 112, -102, -10} : YuvMatrix{66, 129, 25, -38, -74, 112, 112, -94, -18};
}
__global__ __launch_bounds__(256) void k_rgb_to_yuv420(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ yuv, int w, int h, long rgb_stride, long yuv_stride, int matrix) {
    const int bx = blockIdx.x*blockDim.x + threadIdx.x, by = blockIdx.y;          // one 2 x 2 block of pixels per thread
    if (bx >= w/2 || by >= h/2) return;
    const YuvMatrix m = yuv_matrix(matrix);
    const uint8_t* frame = rgb + (long)blockIdx.z*rgb_stride;
    uint8_t* out = yuv + (long)blockIdx.z*yuv_stride;
    int sum_r = 0, sum_g = 0, sum_b = 0;
#pragma unroll
    for (int y = 0; y < 2; y++) {
        const uint8_t* p = frame + ((long)(2*by + y)*w + 2*bx)*3;
        uint8_t luma[2];
#pragma unroll
        for (int x = 0; x < 2; x++) {
            const int r = p[3*x], g = p[3*x + 1], b = p[3*x + 2];
            sum_r += r; sum_g += g; sum_b += b;
            luma[x] = (uint8_t)(((m.yr*r + m.yg*g + m.yb*b + 128) >> 8) + 16);
        }
        *(uchar2*)(out + (long)(2*by + y)*w + 2*bx) = make_uchar2(luma[0], luma[1]);
    }
    const int r = (sum_r + 2) >> 2, g = (sum_g + 2) >> 2, b = (sum_b + 2) >> 2;
    uint8_t* u_plane = out + (long)w*h;
    uint8_t* v_plane = u_plane + (long)(w/2)*(h/2);
    u_plane[(long)by*(w/2) + bx] = (uint8_t)(((m.ur*r + m.ug*g + m.ub*b + 128) >> 8) + 128);
    v_plane[(long)by*(w/2) + bx] = (uint8_t)(((m.vr*r + m.vg*g + m.vb*b + 128) >> 8) + 128);
}

extern "C" int sfx_rgb_to_yuv420(sfx_handle h, const void* rgb, void* yuv, int width, int height, int frames, int matrix) {
    CTX_OR_FAIL(c, h);
    if (!rgb || !yuv || width < 2 || height < 2 || (width & 1) || (height & 1) || frames < 1 || matrix < 0 || matrix > 1)
        return fail(SFX_E_INVALID, "rgb → yuv420p of %d frame(s) of %dx%d (even extents only), matrix %d", frames, width, height, matrix);
    USE_DEVICE(c);
    hipLaunchKernelGGL(k_rgb_to_yuv420, dim3((width/2 + 255)/256, height/2, frames), dim3(256), 0, c->stream, (const uint8_t*)rgb, (uint8_t*)yuv, width, height,
                       (long)width*height*3, (long)width*height*3/2, matrix);
    return launch_status();
}

#include "shm_ring.inc"
#include "flac.inc"

// ---------------------------------------------------------------------------------------------------------
// Audio

struct Audio : Object {
    Context* ctx;
    float* pcm = nullptr;            // planar [channels][samples]
    long samples; int channels, samplerate;
};

extern "C" int sfx_audio_upload(sfx_handle h, const float* interleaved, int64_t samples, int channels, int samplerate, sfx_handle* out) {
    CTX_OR_FAIL(c, h);
    if (!out || samples < 0 || channels < 1 || channels > 8 || (samples > 0 && !interleaved)) return fail(SFX_E_INVALID, "audio of %lld samples x %d channels", (long long)samples, channels);
    USE_DEVICE(c);
    Audio* a = new Audio();
    a->magic = MAGIC_AUDIO; a->ctx = c; a->samples = samples; a->channels = channels; a->samplerate = samplerate;
    std::vector<float> planar((size_t)samples*channels);
    for (int64_t i = 0; i < samples; i++) for (int ch = 0; ch < channels; ch++) planar[(size_t)ch*samples + i] = interleaved[i*channels + ch];
    HIP_TRY(hipMalloc(&a->pcm, planar.size()*sizeof(float) + 16));
    if (!planar.empty()) HIP_TRY(hipMemcpy(a->pcm, planar.data(), planar.size()*sizeof(float), hipMemcpyHostToDevice));
    *out = handle_of(a);
    return SFX_OK;
}

extern "C" int sfx_audio_destroy(sfx_handle h) {
    Audio* a = get<Audio>(h, MAGIC_AUDIO);
    if (!a) return fail(SFX_E_INVALID, "invalid audio handle");
    hipSetDevice(a->ctx->device);
    hipStreamSynchronize(a->ctx->stream);
    hipFree(a->pcm);
    a->magic = 0;
    delete a;
    return SFX_OK;
}

// k-split partial sums of the MFMA filterbank: one per USER of a plan — the plan's own for the per-frame entry points on the context's
// stream, one per tape for its builds on the tape's audio stream — so that a build never shares scratch with a launch on another
// stream (the plan-level buffer was written by both)
struct FilterbankScratch { float* d_partial = nullptr; size_t floats = 0; };

struct Plan : Object {
    Context* ctx;
    int fft_n, window, bins, channels, fft_bins, nnz;
    int fft_size = 0;                // inputs of the transform: 2**fft_n, or int(2**fft_n * sample_rateio) (spectrogram.py:144-146)
    ResampleTap* d_taps = nullptr;   // sample_rateio != 1: where libsamplerate's linear converter reads input sample n of the transform
    int amplitude = 0;               // FourierMagnitude: 0 Power, 1 Amplitude (spectrogram.py:20-26)
    double* d_window = nullptr; double2* d_twiddle = nullptr;
    int *d_indptr = nullptr, *d_indices = nullptr; float* d_data = nullptr;
    float* d_dense = nullptr; int2* d_band = nullptr; int k_pad = 0, row_tiles = 0;
    // scratch that grows on demand
    long* d_tell = nullptr; float* d_power = nullptr; float* d_out = nullptr; int cap_frames = 0;
    FilterbankScratch scratch;                                    // … of the per-frame entry points (the context's stream)
};

static int plan_reserve(Plan* p, int frames) {
    if (frames <= p->cap_frames) return SFX_OK;
    hipStreamSynchronize(p->ctx->stream);
    hipFree(p->d_tell); hipFree(p->d_power); hipFree(p->d_out);
    p->d_tell = nullptr; p->d_power = nullptr; p->d_out = nullptr; p->cap_frames = 0;
    HIP_TRY(hipMalloc(&p->d_tell, sizeof(long)*frames));
    HIP_TRY(hipMalloc(&p->d_power, sizeof(float)*(size_t)frames*p->channels*p->fft_bins));
    HIP_TRY(hipMalloc(&p->d_out, sizeof(float)*(size_t)frames*p->channels*p->bins));
    p->cap_frames = frames;
    return SFX_OK;
}

// A window of the caller's own (spectrogram.py:155-171 multiplies by whatever `self.window(N)` returns, in float64): replaces the
// plan's table; `n` must be the plan's transform size (2**fft_n, or int(2**fft_n * sample_rateio) of a resampled plan).
extern "C" int sfx_stft_plan_window(sfx_handle h, const double* window, int n) {
    Plan* p = get<Plan>(h, MAGIC_PLAN);
    if (!p || !window || n != p->fft_size) return fail(SFX_E_INVALID, "stft plan window: %d values for a plan of %d", n, p ? p->fft_size : 0);
    USE_DEVICE(p->ctx);
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    HIP_TRY(hipMemcpy(p->d_window, window, sizeof(double)*n, hipMemcpyHostToDevice));
    return SFX_OK;
}

static int make_stft_plan(sfx_handle h, int fft_n, int fft_size, const int32_t* tap_a, const int32_t* tap_b, const double* tap_w, int window, int bins, int channels,
                          const int32_t* indptr, const int32_t* indices, const float* data, sfx_handle* out);
extern "C" int sfx_stft_plan(sfx_handle h, int fft_n, int window, int bins, int channels,
                             const int32_t* indptr, const int32_t* indices, const float* data, sfx_handle* out) {
    return make_stft_plan(h, fft_n, (fft_n >= 0 && fft_n < 30) ? (1 << fft_n) : 0, nullptr, nullptr, nullptr, window, bins, channels, indptr, indices, data, out);
}
// `sample_rateio != 1` (spectrogram.py:144-167): the transform takes `fft_size` = int(2**fft_n * ratio) samples, sample n of which is
// (float)(in[tap_a[n]] + tap_w[n]*(in[tap_b[n]] - in[tap_a[n]])) over the last 2**fft_n samples of the ring — the read positions of
// libsamplerate's "linear" converter (samplerate.resample(x, ratio, 'linear'), spectrogram.py:167), which the host derives once per plan
// by running the converter's own float64 position loop (shaderflow_amd/audio/spectrogram.py linear_resample_taps). The built-in windows
// are evaluated for `fft_size`; the filterbank's columns are its fft_size/2 + 1 bins. Power-of-two sizes keep the radix-2 kernel, any
// other size (<= 16 384) takes the float64 DFT sum.
extern "C" int sfx_stft_plan_resampled(sfx_handle h, int fft_n, int fft_size, const int32_t* tap_a, const int32_t* tap_b, const double* tap_w,
                                       int window, int bins, int channels, const int32_t* indptr, const int32_t* indices, const float* data, sfx_handle* out) {
    if (!tap_a || !tap_b || !tap_w) return fail(SFX_E_INVALID, "resampled stft plan: null tap tables");
    return make_stft_plan(h, fft_n, fft_size, tap_a, tap_b, tap_w, window, bins, channels, indptr, indices, data, out);
}
static int make_stft_plan(sfx_handle h, int fft_n, int fft_size, const int32_t* tap_a, const int32_t* tap_b, const double* tap_w, int window, int bins, int channels,
                          const int32_t* indptr, const int32_t* indices, const float* data, sfx_handle* out) {
    CTX_OR_FAIL(c, h);
    if (!out || fft_n < 4 || fft_n > 14 || bins < 1 || channels < 1 || !indptr) return fail(SFX_E_INVALID, "stft plan fft_n=%d bins=%d channels=%d", fft_n, bins, channels);
    if (window < 0 || window > SFX_WINDOW_NONE) return fail(SFX_E_INVALID, "window %d", window);
    if (fft_size < 16 || fft_size > 16384 || (fft_size & 1)) return fail(SFX_E_UNSUPPORTED, "stft transform of %d samples: even sizes from 16 to 16384", fft_size);
    USE_DEVICE(c);
    const int in_size = 1 << fft_n;
    const int N = fft_size, fft_bins = N/2 + 1, nnz = indptr[bins];
    const bool radix2 = (N & (N - 1)) == 0;
    for (int r = 0; r < bins; r++) if (indptr[r] > indptr[r + 1]) return fail(SFX_E_INVALID, "indptr not monotone");
    for (int j = 0; j < nnz; j++) if (indices[j] < 0 || indices[j] >= fft_bins) return fail(SFX_E_INVALID, "column %d outside %d fft bins", indices[j], fft_bins);
    if (tap_a) for (int n = 0; n < N; n++) if (tap_a[n] < 0 || tap_a[n] >= in_size || tap_b[n] < 0 || tap_b[n] >= in_size) return fail(SFX_E_INVALID, "resample tap %d reads outside the %d ring samples", n, in_size);
    Plan* p = new Plan();
    p->magic = MAGIC_PLAN; p->ctx = c; p->fft_n = fft_n; p->fft_size = N; p->window = window; p->bins = bins; p->channels = channels;
    p->fft_bins = fft_bins; p->nnz = nnz;
    // windows: spectrogram.py:92-108 (np.hanning is the symmetric Hann)
    std::vector<double> win(N);
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < N; i++) {
        if (window == SFX_WINDOW_HANNING) win[i] = (N == 1) ? 1.0 : 0.5 + 0.5*::cos(pi*(double)(2*i + 1 - N)/(double)(N - 1));
        else if (window == SFX_WINDOW_HANN_POISSON) win[i] = 0.5*(1.0 - ::cos(2.0*pi*(double)i/(double)N))*::exp(-2.0*::fabs((double)(N - 2*i))/(double)N);
        else win[i] = 1.0;
    }
    // radix-2: exp(-2 pi i k/N) for k < N/2; the DFT sum walks the whole circle
    const int ntw = radix2 ? N/2 : N;
    std::vector<double2> tw(ntw);
    for (int k = 0; k < ntw; k++) { const double ang = -2.0*pi*(double)k/(double)N; tw[k] = make_double2(::cos(ang), ::sin(ang)); }
    HIP_TRY(hipMalloc(&p->d_window, sizeof(double)*N));
    HIP_TRY(hipMalloc(&p->d_twiddle, sizeof(double2)*ntw));
    HIP_TRY(hipMemcpy(p->d_window, win.data(), sizeof(double)*N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->d_twiddle, tw.data(), sizeof(double2)*ntw, hipMemcpyHostToDevice));
    if (tap_a) {
        std::vector<ResampleTap> taps(N);
        for (int n = 0; n < N; n++) taps[n] = ResampleTap{tap_a[n], tap_b[n], tap_w[n]};
        HIP_TRY(hipMalloc(&p->d_taps, sizeof(ResampleTap)*N));
        HIP_TRY(hipMemcpy(p->d_taps, taps.data(), sizeof(ResampleTap)*N, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&p->d_indptr, sizeof(int)*(bins + 1)));
    HIP_TRY(hipMalloc(&p->d_indices, sizeof(int)*(nnz + 1)));
    HIP_TRY(hipMalloc(&p->d_data, sizeof(float)*(nnz + 1)));
    HIP_TRY(hipMemcpy(p->d_indptr, indptr, sizeof(int)*(bins + 1), hipMemcpyHostToDevice));
    if (nnz) {
        HIP_TRY(hipMemcpy(p->d_indices, indices, sizeof(int)*nnz, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_data, data, sizeof(float)*nnz, hipMemcpyHostToDevice));
    }
    // dense banded copy for the MFMA path: rows padded to 32, k padded to 32, per-row-tile k range
    p->row_tiles = (bins + 31)/32;
    p->k_pad = ((fft_bins + 31)/32)*32;
    std::vector<float> dense((size_t)p->row_tiles*32*p->k_pad, 0.0f);
    std::vector<int2> band(p->row_tiles);
    for (int t = 0; t < p->row_tiles; t++) {
        int lo = p->k_pad, hi = 0;
        for (int r = t*32; r < bins && r < t*32 + 32; r++)
            for (int j = indptr[r]; j < indptr[r + 1]; j++) {
                dense[(size_t)r*p->k_pad + indices[j]] = data[j];
                lo = indices[j] < lo ? indices[j] : lo; hi = indices[j] + 1 > hi ? indices[j] + 1 : hi;
            }
        if (hi <= lo) { lo = 0; hi = 0; }
        band[t] = make_int2((lo/32)*32, ((hi + 31)/32)*32);
    }
    HIP_TRY(hipMalloc(&p->d_dense, sizeof(float)*dense.size()));
    HIP_TRY(hipMalloc(&p->d_band, sizeof(int2)*band.size()));
    HIP_TRY(hipMemcpy(p->d_dense, dense.data(), sizeof(float)*dense.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->d_band, band.data(), sizeof(int2)*band.size(), hipMemcpyHostToDevice));
    if (radix2 && (size_t)(N/2)*sizeof(double2) > 64*1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_stft_power, hipFuncAttributeMaxDynamicSharedMemorySize, (N/2)*(int)sizeof(double2)));
    if (!radix2 && (size_t)N*sizeof(double) > 64*1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_dft_power, hipFuncAttributeMaxDynamicSharedMemorySize, N*(int)sizeof(double)));
    *out = handle_of(p);
    return SFX_OK;
}

extern "C" int sfx_stft_plan_magnitude(sfx_handle h, int magnitude) {
    Plan* p = get<Plan>(h, MAGIC_PLAN);
    if (!p) return fail(SFX_E_INVALID, "invalid plan handle");
    if (magnitude != SFX_MAGNITUDE_POWER && magnitude != SFX_MAGNITUDE_AMPLITUDE) return fail(SFX_E_INVALID, "magnitude %d", magnitude);
    p->amplitude = (magnitude == SFX_MAGNITUDE_AMPLITUDE);
    return SFX_OK;
}

extern "C" int sfx_stft_plan_destroy(sfx_handle h) {
    Plan* p = get<Plan>(h, MAGIC_PLAN);
    if (!p) return fail(SFX_E_INVALID, "invalid plan handle");
    hipSetDevice(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
    hipFree(p->d_taps); hipFree(p->d_window); hipFree(p->d_twiddle); hipFree(p->d_indptr); hipFree(p->d_indices); hipFree(p->d_data);
    hipFree(p->d_dense); hipFree(p->d_band); hipFree(p->d_tell); hipFree(p->d_power); hipFree(p->d_out); hipFree(p->scratch.d_partial);
    p->magic = 0;
    delete p;
    return SFX_OK;
}

static int check_audio(const Plan* p, const Audio* a) {
    if (!p || !a) return fail(SFX_E_INVALID, "invalid plan or audio handle");
    if (p->ctx != a->ctx) return fail(SFX_E_INVALID, "plan and audio belong to different contexts");
    if (a->channels != p->channels) return fail(SFX_E_INVALID, "plan built for %d channels, audio has %d (spectrogram.py:306 hard-codes the reshape)", p->channels, a->channels);
    return SFX_OK;
}

// K3: one wave for up to 256 values plus one for the float64 systems (no barriers), 1024 threads for up to 2048 — and, since
// `spectrogram_bins` is anything the user says (spectrogram.py:184; 1 025 stereo bins already exceed 2 048 values), 4 / 8 / 16 values
// per thread for up to 16 384: the early-out's maximum is still ONE block-wide reduction per frame (the recurrence couples the values
// through it, so the scan stays in one block; at 16 waves a thread may hold 128 registers)
constexpr int DYNAMICS_SCAN_LIMIT = 16384;
template <class... Args> static void launch_dynamics_scan(hipStream_t s, int nframes, int n, Args... args) {
    if (n <= 256) hipLaunchKernelGGL((k_dynamics_scan<64, 4, true>), dim3(1), dim3(128), 0, s, nframes, n, args...);
    else if (n <= 2048) hipLaunchKernelGGL((k_dynamics_scan<1024, 2>), dim3(1), dim3(1024), 0, s, nframes, n, args...);
    else if (n <= 4096) hipLaunchKernelGGL((k_dynamics_scan<1024, 4>), dim3(1), dim3(1024), 0, s, nframes, n, args...);
    else if (n <= 8192) hipLaunchKernelGGL((k_dynamics_scan<1024, 8>), dim3(1), dim3(1024), 0, s, nframes, n, args...);
    else hipLaunchKernelGGL((k_dynamics_scan<1024, 16>), dim3(1), dim3(1024), 0, s, nframes, n, args...);
}

// device-side launches shared by the per-frame entry points and the tape
// `what`: 0 power, 1 amplitude (float32 into d_power), 2 the complex spectrum (float64 pairs into d_power, which then is a double2 buffer)
static void launch_stft(const Plan* p, const Audio* a, const long* d_tell, int frames, float* d_power, hipStream_t s, int what = -1) {
    const int N = p->fft_size, in_size = 1 << p->fft_n;
    if (what < 0) what = p->amplitude;
    if ((N & (N - 1)) == 0)
        hipLaunchKernelGGL(k_stft_power, dim3(frames, p->channels), dim3(256), (N/2)*sizeof(double2), s,
                           a->pcm, a->samples, d_tell, __builtin_ctz((unsigned)N), in_size, p->d_taps, p->d_window, p->d_twiddle, d_power, what);
    else
        hipLaunchKernelGGL(k_dft_power, dim3(frames, p->channels), dim3(256), (size_t)N*sizeof(double), s,
                           a->pcm, a->samples, d_tell, N, in_size, p->d_taps, p->d_window, p->d_twiddle, d_power, what);
}
static void launch_filterbank(Plan* p, FilterbankScratch& scratch, int frames, int use_mfma, const float* d_power, float* d_out, hipStream_t s) {
    const int ncols = frames*p->channels;
    const size_t partial = (size_t)FILTERBANK_SPLITS*p->row_tiles*32*ncols;
    if (use_mfma && scratch.floats < partial) {
        hipStreamSynchronize(s);                                      // the scratch's only user is this stream
        hipFree(scratch.d_partial); scratch.d_partial = nullptr; scratch.floats = 0;
        if (hipMalloc(&scratch.d_partial, partial*sizeof(float)) == hipSuccess) scratch.floats = partial;
        else { (void)hipGetLastError(); use_mfma = 0; }               // out of memory for the scratch: the CSR kernel needs none
    }
    if (use_mfma) {
        hipLaunchKernelGGL(k_filterbank_mfma, dim3((ncols + 31)/32, p->row_tiles, FILTERBANK_SPLITS), dim3(64), 0, s,
                           p->d_dense, p->k_pad, p->d_band, p->fft_bins, ncols, d_power, scratch.d_partial);
        hipLaunchKernelGGL(k_filterbank_reduce, dim3((ncols + 31)/32, (p->bins + 7)/8), dim3(256), 0, s, scratch.d_partial, p->row_tiles*32, p->bins, p->channels, ncols, d_out);
    } else {
        const long total = (long)ncols*p->bins;
        hipLaunchKernelGGL(k_filterbank_csr, dim3((unsigned)((total + 255)/256)), dim3(256), 0, s,
                           p->d_indptr, p->d_indices, p->d_data, p->bins, p->channels, p->fft_bins, ncols, d_power, d_out);
    }
}

extern "C" int sfx_stft_power(sfx_handle hp, sfx_handle ha, const int64_t* tell, int nframes, float* power) {
    Plan* p = get<Plan>(hp, MAGIC_PLAN); Audio* a = get<Audio>(ha, MAGIC_AUDIO);
    int rc = check_audio(p, a);
    if (rc) return rc;
    if (!tell || !power || nframes < 1) return fail(SFX_E_INVALID, "null tell/power or no frames");
    USE_DEVICE(p->ctx);
    if ((rc = plan_reserve(p, nframes))) return rc;
    hipStream_t s = p->ctx->stream;
    HIP_TRY(hipMemcpyAsync(p->d_tell, tell, sizeof(long)*nframes, hipMemcpyHostToDevice, s));
    launch_stft(p, a, p->d_tell, nframes, p->d_power, s);
    if ((rc = launch_status())) return rc;
    HIP_TRY(hipMemcpyAsync(power, p->d_power, sizeof(float)*(size_t)nframes*p->channels*p->fft_bins, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SFX_OK;
}

// `magnitude` callables of the user's own (spectrogram.py:20-41, 169-171 accepts ANY callable on the complex spectrum): the device computes
// np.fft.rfft(window*frame) and hands the float64 pairs over, the host applies the callable, sfx_filterbank_apply takes its float32 result
// through the filterbank. A slow path (two host round trips per call) for an option nothing in the reference's tree uses — but it works.
extern "C" int sfx_stft_spectrum(sfx_handle hp, sfx_handle ha, const int64_t* tell, int nframes, double* spectrum) {
    Plan* p = get<Plan>(hp, MAGIC_PLAN); Audio* a = get<Audio>(ha, MAGIC_AUDIO);
    int rc = check_audio(p, a);
    if (rc) return rc;
    if (!tell || !spectrum || nframes < 1) return fail(SFX_E_INVALID, "null tell/spectrum or no frames");
    USE_DEVICE(p->ctx);
    if ((rc = plan_reserve(p, nframes))) return rc;
    hipStream_t s = p->ctx->stream;
    const size_t bytes = sizeof(double)*2*(size_t)nframes*p->channels*p->fft_bins;
    void* d_spectrum = nullptr;
    HIP_TRY(hipMalloc(&d_spectrum, bytes));
    hipError_t e = hipMemcpyAsync(p->d_tell, tell, sizeof(long)*nframes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) { launch_stft(p, a, p->d_tell, nframes, (float*)d_spectrum, s, 2); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(spectrum, d_spectrum, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    hipFree(d_spectrum);
    return e == hipSuccess ? SFX_OK : fail(SFX_E_HIP, "stft spectrum: %s", hipGetErrorString(e));
}

extern "C" int sfx_filterbank_apply(sfx_handle hp, const float* magnitudes, int nframes, int use_mfma, float* out) {
    Plan* p = get<Plan>(hp, MAGIC_PLAN);
    if (!p) return fail(SFX_E_INVALID, "invalid plan handle");
    if (!magnitudes || !out || nframes < 1) return fail(SFX_E_INVALID, "null magnitudes/out or no frames");
    USE_DEVICE(p->ctx);
    int rc = plan_reserve(p, nframes);
    if (rc) return rc;
    hipStream_t s = p->ctx->stream;
    HIP_TRY(hipMemcpyAsync(p->d_power, magnitudes, sizeof(float)*(size_t)nframes*p->channels*p->fft_bins, hipMemcpyHostToDevice, s));
    launch_filterbank(p, p->scratch, nframes, use_mfma, p->d_power, p->d_out, s);
    if ((rc = launch_status())) return rc;
    HIP_TRY(hipMemcpyAsync(out, p->d_out, sizeof(float)*(size_t)nframes*p->channels*p->bins, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SFX_OK;
}

extern "C" int sfx_spectrogram_targets(sfx_handle hp, sfx_handle ha, const int64_t* tell, int nframes, int use_mfma, float* out) {
    Plan* p = get<Plan>(hp, MAGIC_PLAN); Audio* a = get<Audio>(ha, MAGIC_AUDIO);
    int rc = check_audio(p, a);
    if (rc) return rc;
    if (!tell || !out || nframes < 1) return fail(SFX_E_INVALID, "null tell/out or no frames");
    USE_DEVICE(p->ctx);
    if ((rc = plan_reserve(p, nframes))) return rc;
    hipStream_t s = p->ctx->stream;
    HIP_TRY(hipMemcpyAsync(p->d_tell, tell, sizeof(long)*nframes, hipMemcpyHostToDevice, s));
    launch_stft(p, a, p->d_tell, nframes, p->d_power, s);
    launch_filterbank(p, p->scratch, nframes, use_mfma, p->d_power, p->d_out, s);
    if ((rc = launch_status())) return rc;
    HIP_TRY(hipMemcpyAsync(out, p->d_out, sizeof(float)*(size_t)nframes*p->channels*p->bins, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SFX_OK;
}

extern "C" int sfx_waveform_rows(sfx_handle ha, const int64_t* tell, int nframes, int chunk, int points, int reducer, float* out) {
    Audio* a = get<Audio>(ha, MAGIC_AUDIO);
    if (!a || !tell || !out || nframes < 1 || chunk < 1 || points < 1) return fail(SFX_E_INVALID, "invalid audio handle or arguments");
    USE_DEVICE(a->ctx);
    hipStream_t s = a->ctx->stream;
    long* d_tell; float* d_rows;
    const size_t n = (size_t)nframes*points*a->channels;
    HIP_TRY(hipMalloc(&d_tell, sizeof(long)*nframes));
    HIP_TRY(hipMalloc(&d_rows, sizeof(float)*n));
    HIP_TRY(hipMemcpyAsync(d_tell, tell, sizeof(long)*nframes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_waveform_rows, dim3((points*a->channels + 3)/4, nframes), dim3(256), 0, s,
                       a->pcm, a->samples, a->channels, d_tell, chunk, points, reducer, d_rows);
    int rc = launch_status();
    if (!rc) { hipMemcpyAsync(out, d_rows, sizeof(float)*n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
    hipFree(d_tell); hipFree(d_rows);
    return rc;
}

extern "C" int sfx_volume_std(sfx_handle ha, const int64_t* tell, int nframes, int window_samples, float* out) {
    Audio* a = get<Audio>(ha, MAGIC_AUDIO);
    if (!a || !tell || !out || nframes < 1 || window_samples < 1) return fail(SFX_E_INVALID, "invalid audio handle or arguments");
    USE_DEVICE(a->ctx);
    hipStream_t s = a->ctx->stream;
    long* d_tell; float* d_out;
    HIP_TRY(hipMalloc(&d_tell, sizeof(long)*nframes));
    HIP_TRY(hipMalloc(&d_out, sizeof(float)*2*nframes));
    HIP_TRY(hipMemcpyAsync(d_tell, tell, sizeof(long)*nframes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_volume_std, dim3(nframes), dim3(256), 0, s, a->pcm, a->samples, a->channels, d_tell, window_samples, d_out);
    int rc = launch_status();
    if (!rc) { hipMemcpyAsync(out, d_out, sizeof(float)*2*nframes, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
    hipFree(d_tell); hipFree(d_out);
    return rc;
}

// DynamicNumber.next over a run of frames, on its own (SURVEY.md §8b last row; dynamics.py:197-250). Host arrays in and out.
extern "C" int sfx_dynamics_scan(sfx_handle h, int nframes, int n, const float* targets, const sfx_dyn_coeff_f32* coeff,
                                 float precision, float* state, float* values) {
    CTX_OR_FAIL(c, h);
    if (nframes < 1 || n < 1 || !targets || !coeff || !state || !values) return fail(SFX_E_INVALID, "dynamics scan: null array or nothing to do");
    if (n > DYNAMICS_SCAN_LIMIT) return fail(SFX_E_UNSUPPORTED, "dynamics scan handles up to %d values per system, got %d", DYNAMICS_SCAN_LIMIT, n);
    USE_DEVICE(c);
    hipStream_t s = c->stream;
    float *d_targets = nullptr, *d_state = nullptr, *d_values = nullptr; DynCoeffF32* d_coeff = nullptr;
    const size_t frame_bytes = sizeof(float)*(size_t)nframes*n;
    const bool ok = hipMalloc(&d_targets, frame_bytes) == hipSuccess && hipMalloc(&d_values, frame_bytes) == hipSuccess &&
                    hipMalloc(&d_state, sizeof(float)*3*n) == hipSuccess && hipMalloc(&d_coeff, sizeof(DynCoeffF32)*nframes) == hipSuccess;
    int rc = ok ? SFX_OK : fail(SFX_E_HIP, "dynamics scan of %d frames x %d values: out of device memory", nframes, n);
    if (!rc) {
        hipMemcpyAsync(d_targets, targets, frame_bytes, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(d_state, state, sizeof(float)*3*n, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(d_coeff, coeff, sizeof(DynCoeffF32)*nframes, hipMemcpyHostToDevice, s);
        launch_dynamics_scan(s, nframes, n, d_targets, d_coeff, precision, d_state, d_values,
                             (const float*)nullptr, (const DynCoeffF64*)nullptr, (const DynCoeffF64*)nullptr, 0.0, 0, 0,
                             (ScalarState*)nullptr, (const FrameClock*)nullptr, (FrameDyn*)nullptr);
        rc = launch_status();
    }
    if (!rc) {
        hipMemcpyAsync(values, d_values, frame_bytes, hipMemcpyDeviceToHost, s);
        hipMemcpyAsync(state, d_state, sizeof(float)*3*n, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) rc = fail(SFX_E_HIP, "dynamics scan: stream synchronisation failed");
    }
    hipFree(d_targets); hipFree(d_values); hipFree(d_state); hipFree(d_coeff);
    return rc;
}

extern "C" int sfx_dynamics_scan_f64(sfx_handle h, int nframes, int nsystems, const double* targets, const sfx_dyn_coeff_f64* coeff,
                                     double precision, int integrate, double* state, double* out) {
    CTX_OR_FAIL(c, h);
    if (nframes < 1 || nsystems < 1 || !targets || !coeff || !state || !out) return fail(SFX_E_INVALID, "dynamics scan: null array or nothing to do");
    static_assert(sizeof(ScalarState) == 4*sizeof(double), "state = value, derivative, previous, integral");
    USE_DEVICE(c);
    hipStream_t s = c->stream;
    double *d_targets = nullptr, *d_out = nullptr; ScalarState* d_state = nullptr; DynCoeffF64* d_coeff = nullptr;
    const size_t count = (size_t)nframes*nsystems;
    const bool ok = hipMalloc(&d_targets, sizeof(double)*count) == hipSuccess && hipMalloc(&d_out, sizeof(double)*3*count) == hipSuccess &&
                    hipMalloc(&d_state, sizeof(ScalarState)*nsystems) == hipSuccess && hipMalloc(&d_coeff, sizeof(DynCoeffF64)*count) == hipSuccess;
    int rc = ok ? SFX_OK : fail(SFX_E_HIP, "dynamics scan of %d frames x %d systems: out of device memory", nframes, nsystems);
    if (!rc) {
        hipMemcpyAsync(d_targets, targets, sizeof(double)*count, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(d_state, state, sizeof(ScalarState)*nsystems, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(d_coeff, coeff, sizeof(DynCoeffF64)*count, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(k_dynamics_scan_f64, dim3((nsystems + 63)/64), dim3(64), 0, s, nframes, nsystems, d_targets, d_coeff, precision, integrate, d_state, d_out);
        rc = launch_status();
    }
    if (!rc) {
        hipMemcpyAsync(out, d_out, sizeof(double)*3*count, hipMemcpyDeviceToHost, s);
        hipMemcpyAsync(state, d_state, sizeof(ScalarState)*nsystems, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) rc = fail(SFX_E_HIP, "dynamics scan: stream synchronisation failed");
    }
    hipFree(d_targets); hipFree(d_out); hipFree(d_state); hipFree(d_coeff);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------
// Tape

// The arrays a batch of frames lives in exist TWICE (two banks): sfx_tape_build fills the bank the last render did not read, on
// the tape's own stream, while the context's stream still renders from the other one — the audio kernels of batch i + 1 (a chain
// of small latency-bound launches, 0.13-0.2 ms) run beside the render of batch i instead of in front of batch i + 1's, and the host
// never waits for a render to hand over its schedule (the borrowed host arrays are copied to pinned memory of the bank).
// Events order the two streams: `built` (recorded after a bank's last audio kernel; renders and reads wait for it), `rendered`
// (recorded after every render from a bank; the build that refills the bank waits for it). The recurrences' state (d_state,
// d_scalars, the scrolling ring) exists once: only the tape's stream touches it, in frame order.
struct TapeBank {
    long* d_tell = nullptr; float *d_power = nullptr, *d_targets = nullptr, *d_columns = nullptr, *d_rows = nullptr, *d_loudness = nullptr;
    FrameDyn* d_dyn = nullptr; DynCoeffF32* d_coeff = nullptr; DynCoeffF64 *d_vol = nullptr, *d_std = nullptr; FrameClock* d_clock = nullptr;
    VisualizerConsts* d_vis = nullptr; float *d_bars = nullptr, *d_scroll = nullptr;
    char* staging = nullptr;         // pinned: the host's schedule arrays of the batch, laid out like d_schedule
    char* d_schedule = nullptr;      // d_tell | d_clock | d_coeff | d_vol | d_std in one allocation: one copy per build
    hipEvent_t built = nullptr, rendered = nullptr;
};
struct Tape : Object {
    Plan* plan; Audio* audio; Context* ctx;
    sfx_tape_desc desc;
    int max_frames, n;               // n = bins*channels
    // the bank the last sfx_tape_build filled (what renders and reads see)
    long* d_tell = nullptr; float *d_power = nullptr, *d_targets = nullptr, *d_columns = nullptr, *d_rows = nullptr, *d_loudness = nullptr;
    FrameDyn* d_dyn = nullptr;
    DynCoeffF32* d_coeff = nullptr; DynCoeffF64 *d_vol = nullptr, *d_std = nullptr; FrameClock* d_clock = nullptr;
    VisualizerConsts* d_vis = nullptr;
    float* d_bars = nullptr;         // sqrt(column/1000) of every frame of the batch (visualizer.frag:45)
    float* d_scroll = nullptr;       // scrolling spectrogram: the texture's state per frame of the batch
    TapeBank bank[2]; int current = 0; bool built_once = false;
    hipStream_t audio_stream = nullptr;
    FilterbankScratch scratch;       // of this tape's builds (audio_stream)
    float* d_state = nullptr; ScalarState* d_scalars = nullptr;
    void* d_screen = nullptr; size_t screen_bytes = 0;   // iScreen scratch of the two-pass path (frames of a batch)
    // scrolling spectrogram (length_samples > 1, spectrogram.py:298-311): ring of the last columns
    int width = 1, ring_frames = 0; long frames_done = 0;
    float* d_ring = nullptr;
};
static void tape_select(Tape* t, int b) {
    const TapeBank& k = t->bank[b];
    t->d_tell = k.d_tell; t->d_power = k.d_power; t->d_targets = k.d_targets; t->d_columns = k.d_columns; t->d_rows = k.d_rows;
    t->d_loudness = k.d_loudness; t->d_dyn = k.d_dyn; t->d_coeff = k.d_coeff; t->d_vol = k.d_vol; t->d_std = k.d_std; t->d_clock = k.d_clock;
    t->d_vis = k.d_vis; t->d_bars = k.d_bars; t->d_scroll = k.d_scroll;
    t->current = b;
}
// offsets of the five schedule arrays in a bank's block (each 16-byte aligned), [5] = the block's size
struct ScheduleLayout { size_t at[6]; };
static ScheduleLayout schedule_layout(int frames) {
    const size_t sizes[5] = {sizeof(long), sizeof(FrameClock), sizeof(DynCoeffF32), sizeof(DynCoeffF64), sizeof(DynCoeffF64)};
    ScheduleLayout l; size_t at = 0;
    for (int i = 0; i < 5; i++) { l.at[i] = at; at += (sizes[i]*(size_t)frames + 15) & ~(size_t)15; }
    l.at[5] = at;
    return l;
}
static size_t tape_staging_bytes(int frames) { return std::max(schedule_layout(frames).at[5], sizeof(FrameDyn)*(size_t)frames); }
// streams, events and pinned staging of both banks; false = out of memory
static bool tape_open_streams(Tape* t) {
    // the audio kernels are small and the render kernel fills the chip: at the highest priority their workgroups take the next free
    // slots instead of queueing behind the render's
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    // (SHADERFLOW_TAPE_PRIORITY=normal: A/B switch for measurements — tools/experiments/timeline_overlap.py)
    const char* priority = getenv("SHADERFLOW_TAPE_PRIORITY");
    if (priority && !strcmp(priority, "normal")) greatest = 0;
    if (hipStreamCreateWithPriority(&t->audio_stream, hipStreamNonBlocking, greatest) != hipSuccess) return false;
    for (TapeBank& k : t->bank) {
        if (hipEventCreateWithFlags(&k.built, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&k.rendered, hipEventDisableTiming) != hipSuccess) return false;
        if (hipHostMalloc((void**)&k.staging, tape_staging_bytes(t->max_frames), hipHostMallocDefault) != hipSuccess) return false;
    }
    return true;
}

extern "C" int sfx_tape_reset(sfx_handle h) {
    Tape* t = get<Tape>(h, MAGIC_TAPE);
    if (!t) return fail(SFX_E_INVALID, "invalid tape handle");
    if (!t->plan) return SFX_OK;                                    // clock tape: no recurrences to reset
    USE_DEVICE(t->ctx);
    HIP_TRY(hipMemsetAsync(t->d_state, 0, sizeof(float)*3*t->n, t->audio_stream));   // in order with the builds before and after it
    HIP_TRY(hipMemsetAsync(t->d_scalars, 0, sizeof(ScalarState)*2, t->audio_stream));
    t->frames_done = 0;                                             // the scrolling texture starts empty again
    return SFX_OK;
}

// A tape without audio (plan == 0 and audio == 0, `ctx_for_clock_tape` says where it lives): only the frame clock varies
// between the frames of a batch — scenes without audio modules (Basic, ShaderToy, RayMarch, the fractals).
extern "C" int sfx_clock_tape_create(sfx_handle hc, int max_frames, sfx_handle* out) {
    CTX_OR_FAIL(c, hc);
    if (!out || max_frames < 1) return fail(SFX_E_INVALID, "null output or no frames");
    USE_DEVICE(c);
    Tape* t = new Tape();
    memset(static_cast<void*>(&t->desc), 0, sizeof t->desc);
    t->magic = MAGIC_TAPE; t->plan = nullptr; t->audio = nullptr; t->ctx = c; t->max_frames = max_frames; t->n = 0;
    bool ok = tape_open_streams(t);
    for (TapeBank& k : t->bank)
        ok = ok && hipMalloc(&k.d_dyn, sizeof(FrameDyn)*max_frames) == hipSuccess && hipMalloc(&k.d_vis, sizeof(VisualizerConsts)*max_frames) == hipSuccess;
    tape_select(t, 0);
    if (!ok) {
        sfx_tape_destroy(handle_of(t));
        return fail(SFX_E_HIP, "clock tape of %d frames: out of device memory", max_frames);
    }
    *out = handle_of(t);
    return SFX_OK;
}

extern "C" int sfx_tape_create(sfx_handle hp, sfx_handle ha, const sfx_tape_desc* desc, int max_frames, sfx_handle* out) {
    Plan* p = get<Plan>(hp, MAGIC_PLAN); Audio* a = get<Audio>(ha, MAGIC_AUDIO);
    int rc = check_audio(p, a);
    if (rc) return rc;
    if (!desc || !out || max_frames < 1) return fail(SFX_E_INVALID, "null desc/output or no frames");
    if (p->bins*p->channels > DYNAMICS_SCAN_LIMIT) return fail(SFX_E_UNSUPPORTED, "dynamics scan handles up to %d spectrogram values, got %d", DYNAMICS_SCAN_LIMIT, p->bins*p->channels);
    USE_DEVICE(p->ctx);
    Tape* t = new Tape();
    t->magic = MAGIC_TAPE; t->plan = p; t->audio = a; t->ctx = p->ctx; t->desc = *desc; t->max_frames = max_frames;
    t->n = p->bins*p->channels;
    const size_t F = max_frames;
    const int pts = desc->points > 0 ? desc->points : 1;
    t->width = desc->length_samples > 1 ? desc->length_samples : 1;
    bool allocated = tape_open_streams(t) &&
        hipMalloc(&t->d_state, sizeof(float)*3*t->n) == hipSuccess &&
        hipMalloc(&t->d_scalars, sizeof(ScalarState)*2) == hipSuccess;
    for (TapeBank& k : t->bank)
        allocated = allocated &&
            hipMalloc((void**)&k.d_schedule, schedule_layout(max_frames).at[5]) == hipSuccess &&
            hipMalloc(&k.d_power, sizeof(float)*F*p->channels*p->fft_bins) == hipSuccess &&
            hipMalloc(&k.d_targets, sizeof(float)*F*t->n) == hipSuccess &&
            hipMalloc(&k.d_columns, sizeof(float)*F*t->n) == hipSuccess &&
            hipMalloc(&k.d_rows, sizeof(float)*F*pts*a->channels) == hipSuccess &&
            hipMalloc(&k.d_loudness, sizeof(float)*F*2) == hipSuccess &&
            hipMalloc(&k.d_dyn, sizeof(FrameDyn)*F) == hipSuccess &&
            hipMalloc(&k.d_vis, sizeof(VisualizerConsts)*F) == hipSuccess &&
            hipMalloc(&k.d_bars, sizeof(float)*F*t->n) == hipSuccess &&
            (t->width <= 1 || hipMalloc(&k.d_scroll, sizeof(float)*F*t->n*t->width) == hipSuccess);
    if (allocated && t->width > 1) {
        t->ring_frames = t->width + max_frames;
        allocated = hipMalloc(&t->d_ring, sizeof(float)*(size_t)t->ring_frames*t->n) == hipSuccess;
    }
    if (allocated) {
        const ScheduleLayout l = schedule_layout(max_frames);
        for (TapeBank& k : t->bank) {
            k.d_tell = (long*)(k.d_schedule + l.at[0]); k.d_clock = (FrameClock*)(k.d_schedule + l.at[1]); k.d_coeff = (DynCoeffF32*)(k.d_schedule + l.at[2]);
            k.d_vol = (DynCoeffF64*)(k.d_schedule + l.at[3]); k.d_std = (DynCoeffF64*)(k.d_schedule + l.at[4]);
        }
    }
    tape_select(t, 0);
    if (!allocated) {
        sfx_tape_destroy(handle_of(t));                             // frees what was allocated (hipFree(nullptr) is a no-op)
        return fail(SFX_E_HIP, "tape of %d frames: out of device memory", max_frames);
    }
    *out = handle_of(t);
    return sfx_tape_reset(*out);
}

extern "C" int sfx_tape_build(sfx_handle h, int nframes, const int64_t* tell, const sfx_frame_clock* clock,
                              const sfx_dyn_coeff_f32* spectrogram, const sfx_dyn_coeff_f64* volume, const sfx_dyn_coeff_f64* std_) {
    Tape* t = get<Tape>(h, MAGIC_TAPE);
    if (!t) return fail(SFX_E_INVALID, "invalid tape handle");
    if (nframes < 1 || nframes > t->max_frames || !clock) return fail(SFX_E_INVALID, "tape build of %d frames (capacity %d) or null clock", nframes, t->max_frames);
    USE_DEVICE(t->ctx);
    // the bank the last render did not read; its previous copy out of the pinned staging is long done (two builds ago) — the wait
    // is there for callers that build without rendering
    const int b = t->built_once ? (t->current ^ 1) : 0;
    TapeBank& k = t->bank[b];
    HIP_TRY(hipEventSynchronize(k.built));
    hipStream_t s = t->audio_stream;
    HIP_TRY(hipStreamWaitEvent(s, k.rendered, 0));                  // the renders that read this bank
    if (!t->plan) {                                                 // clock tape: the per-frame uniforms are the clock itself
        FrameDyn* dyn = (FrameDyn*)k.staging;
        for (int f = 0; f < nframes; f++) {
            memset(&dyn[f], 0, sizeof(FrameDyn));
            dyn[f].iTime = clock[f].iTime; dyn[f].iTau = clock[f].iTau; dyn[f].iSpectrogramOffset = clock[f].iSpectrogramOffset; dyn[f].iFrame = clock[f].iFrame;
        }
        HIP_TRY(hipMemcpyAsync(k.d_dyn, dyn, sizeof(FrameDyn)*nframes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(k.built, s));
        tape_select(t, b); t->built_once = true;
        return SFX_OK;
    }
    if (!tell || !spectrogram || !volume || !std_) return fail(SFX_E_INVALID, "tape build with null audio schedule arrays");
    static_assert(sizeof(sfx_dyn_coeff_f32) == sizeof(DynCoeffF32) && sizeof(sfx_dyn_coeff_f64) == sizeof(DynCoeffF64) && sizeof(sfx_frame_clock) == sizeof(FrameClock), "ABI structs");
    static_assert(sizeof(long) == sizeof(int64_t), "tell");
    Plan* p = t->plan; const Audio* a = t->audio;
    // host arrays are borrowed for the call only: into the bank's pinned staging, from there to the device in ONE copy behind the
    // host's back
    const ScheduleLayout l = schedule_layout(t->max_frames);
    memcpy(k.staging + l.at[0], tell, sizeof(long)*nframes);
    memcpy(k.staging + l.at[1], clock, sizeof(FrameClock)*nframes);
    memcpy(k.staging + l.at[2], spectrogram, sizeof(DynCoeffF32)*nframes);
    memcpy(k.staging + l.at[3], volume, sizeof(DynCoeffF64)*nframes);
    memcpy(k.staging + l.at[4], std_, sizeof(DynCoeffF64)*nframes);
    HIP_TRY(hipMemcpyAsync(k.d_schedule, k.staging, l.at[4] + sizeof(DynCoeffF64)*nframes, hipMemcpyHostToDevice, s));
    launch_stft(p, a, k.d_tell, nframes, k.d_power, s);
    launch_filterbank(p, t->scratch, nframes, t->desc.use_mfma, k.d_power, k.d_targets, s);
    if (t->desc.points > 0)
        hipLaunchKernelGGL(k_waveform_rows, dim3((t->desc.points*a->channels + 3)/4, nframes), dim3(256), 0, s,
                           a->pcm, a->samples, a->channels, k.d_tell, t->desc.chunk_size, t->desc.points, t->desc.reducer, k.d_rows);
    hipLaunchKernelGGL(k_volume_std, dim3(nframes), dim3(256), 0, s, a->pcm, a->samples, a->channels, k.d_tell, t->desc.volume_window, k.d_loudness);
    launch_dynamics_scan(s, nframes, t->n, k.d_targets, k.d_coeff, (float)t->desc.precision,
                         t->d_state, k.d_columns, k.d_loudness, k.d_vol, k.d_std, t->desc.precision,
                         t->desc.volume_integrate, t->desc.std_integrate, t->d_scalars, k.d_clock, k.d_dyn);
    if (t->width > 1) {
        const long count = (long)nframes*t->n;
        hipLaunchKernelGGL(k_spectrogram_ring_store, dim3((unsigned)((count + 255)/256)), dim3(256), 0, s, k.d_columns, nframes, t->n, t->frames_done, t->ring_frames, t->d_ring);
        const long texels = count*t->width;
        hipLaunchKernelGGL(k_spectrogram_scroll, dim3((unsigned)((texels + 255)/256)), dim3(256), 0, s, t->d_ring, t->ring_frames, t->frames_done, nframes,
                           p->bins, p->channels, t->width, k.d_scroll);
    }
    t->frames_done += nframes;
    const int rc = launch_status();
    HIP_TRY(hipEventRecord(k.built, s));
    tape_select(t, b); t->built_once = true;
    return rc;
}

extern "C" int sfx_tape_read(sfx_handle h, int what, int frame0, int nframes, void* out, size_t nbytes) {
    Tape* t = get<Tape>(h, MAGIC_TAPE);
    if (!t || !out) return fail(SFX_E_INVALID, "invalid tape handle or output");
    if (frame0 < 0 || nframes < 1 || frame0 + nframes > t->max_frames) return fail(SFX_E_INVALID, "frames [%d, %d) outside the tape", frame0, frame0 + nframes);
    USE_DEVICE(t->ctx);
    const char* src; size_t per;
    const int pts = t->desc.points > 0 ? t->desc.points : 1;
    if (!t->plan && what != SFX_TAPE_UNIFORMS) return fail(SFX_E_INVALID, "a clock tape holds the per-frame uniforms only");
    switch (what) {
        case SFX_TAPE_SPECTROGRAM: src = (const char*)t->d_columns; per = sizeof(float)*t->n; break;
        case SFX_TAPE_WAVEFORM: src = (const char*)t->d_rows; per = sizeof(float)*pts*t->audio->channels; break;
        case SFX_TAPE_UNIFORMS: src = (const char*)t->d_dyn; per = sizeof(FrameDyn); break;
        case SFX_TAPE_TARGETS: src = (const char*)t->d_targets; per = sizeof(float)*t->n; break;
        case SFX_TAPE_LOUDNESS: src = (const char*)t->d_loudness; per = sizeof(float)*2; break;
        case SFX_TAPE_SCROLL:
            if (t->width <= 1) return fail(SFX_E_INVALID, "the tape has no scrolling spectrogram (length_samples <= 1)");
            src = (const char*)t->d_scroll; per = sizeof(float)*t->n*t->width; break;
        default: return fail(SFX_E_INVALID, "tape section %d", what);
    }
    if (nbytes != per*nframes) return fail(SFX_E_INVALID, "tape read of %zu bytes, section needs %zu", nbytes, per*nframes);
    HIP_TRY(hipStreamWaitEvent(t->ctx->stream, t->bank[t->current].built, 0));
    HIP_TRY(hipMemcpyAsync(out, src + per*frame0, nbytes, hipMemcpyDeviceToHost, t->ctx->stream));
    HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    return SFX_OK;
}

extern "C" int sfx_tape_destroy(sfx_handle h) {
    Tape* t = get<Tape>(h, MAGIC_TAPE);
    if (!t) return fail(SFX_E_INVALID, "invalid tape handle");
    hipSetDevice(t->ctx->device);
    if (t->audio_stream) hipStreamSynchronize(t->audio_stream);
    hipStreamSynchronize(t->ctx->stream);
    for (TapeBank& k : t->bank) {
        hipFree(k.d_schedule); hipFree(k.d_power); hipFree(k.d_targets); hipFree(k.d_columns); hipFree(k.d_rows); hipFree(k.d_loudness);
        hipFree(k.d_dyn); hipFree(k.d_vis); hipFree(k.d_bars); hipFree(k.d_scroll);
        if (k.staging) hipHostFree(k.staging);
        if (k.built) hipEventDestroy(k.built);
        if (k.rendered) hipEventDestroy(k.rendered);
    }
    hipFree(t->d_state); hipFree(t->d_scalars); hipFree(t->d_screen); hipFree(t->d_ring); hipFree(t->scratch.d_partial);
    if (t->audio_stream) hipStreamDestroy(t->audio_stream);
    t->magic = 0;
    delete t;
    return SFX_OK;
}

extern "C" int sfx_render_tape(sfx_handle hp, sfx_handle ht, int frame0, int nframes, int width, int height,
                               int ssaa_x1000, int subsample, void* device_out) {
    Program* p = get<Program>(hp, MAGIC_PROG);
    Tape* t = get<Tape>(ht, MAGIC_TAPE);
    if (!p || !t || !device_out) return fail(SFX_E_INVALID, "invalid program/tape handle or output");
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
    a.dyn = t->d_dyn; a.frame0 = frame0;
    // the bank was filled on the tape's stream: this stream waits for its last audio kernel, and leaves a mark behind its own last
    // kernel that the build refilling the bank will wait for (Tape, above)
    HIP_TRY(hipStreamWaitEvent(p->ctx->stream, t->bank[t->current].built, 0));
    struct RenderedMark { hipEvent_t event; hipStream_t stream; ~RenderedMark() { hipEventRecord(event, stream); } } mark{t->bank[t->current].rendered, p->ctx->stream};
    if (t->plan) {
    // iSpectrogram: width 1 (length=0 scenes), height bins, RG32F (spectrogram.py:298-311); the bound texture's
    // sampler state is kept, only its storage is redirected to the tape column of the frame
    // (length > 0: the texture is `width` columns wide and every frame of the batch gets its own state of it, k_spectrogram_scroll)
    const bool scrolling = (t->width > 1);
    a.tape_spectrogram = scrolling ? t->d_scroll : t->d_columns; a.spectrogram_stride = (long)t->n*t->width;
    if (!a.tex[TEX_SPECTROGRAM].data) {
        Tex& s = a.tex[TEX_SPECTROGRAM];
        s.width = t->width; s.height = t->plan->bins; s.components = t->plan->channels; s.dtype = DT_F32; s.filter = FILTER_NEAREST; s.repeat_x = 1; s.repeat_y = 0;
    }
    if (a.tex[TEX_SPECTROGRAM].width != t->width || a.tex[TEX_SPECTROGRAM].height != t->plan->bins || a.tex[TEX_SPECTROGRAM].components != t->plan->channels)
        return fail(SFX_E_INVALID, "the bound iSpectrogram is %d x %d x %d, the tape was created for %d x %d x %d (length_samples x bins x channels)",
                    a.tex[TEX_SPECTROGRAM].width, a.tex[TEX_SPECTROGRAM].height, a.tex[TEX_SPECTROGRAM].components, t->width, t->plan->bins, t->plan->channels);
    if (t->desc.points > 0) {
        a.tape_waveform = t->d_rows; a.waveform_stride = (long)t->desc.points*t->audio->channels;
        if (!a.tex[TEX_WAVEFORM].data) {
            Tex& w = a.tex[TEX_WAVEFORM];
            w.width = t->desc.points; w.height = 1; w.components = t->audio->channels; w.dtype = DT_F32; w.filter = FILTER_LINEAR; w.repeat_x = 0; w.repeat_y = 0;
        }
    }
    // placeholders so that texel() sees a non-null base before frame_view() redirects it
    a.tex[TEX_SPECTROGRAM].data = a.tape_spectrogram;
    if (t->desc.points > 0) a.tex[TEX_WAVEFORM].data = t->d_rows;
    }
    int rc = check_samplers(p->fragment, a);
    if (rc) return rc;
    a.has_vis = 0;                                                  // per-frame audio uniforms live on the device
    if (p->fragment == FRAG_VISUALIZER) {
        sfl::visualizer_consts_frames(t->d_dyn, frame0, nframes, t->d_vis, p->ctx->stream);
        a.vis_consts = t->d_vis;
        if (t->plan && t->width == 1 && a.tex[TEX_SPECTROGRAM].components == 2 && a.tex[TEX_SPECTROGRAM].filter == FILTER_NEAREST) {
            const long count = (long)nframes*t->n;
            sfl::visualizer_bars(t->d_columns + (long)frame0*t->n, count, t->d_bars + (long)frame0*t->n, p->ctx->stream);
            a.tape_bars = t->d_bars;
        }
    }
    if (fused) {
        if ((rc = launch_fused_p(p, a, ssaa, nframes, p->ctx->stream))) return rc;
        return launch_status();
    }
    // two passes, batched: the fragment into an RGBA8 iScreen scratch per frame, then final.glsl (shader.py:388-405)
    const size_t screen_frame = (size_t)a.wr*a.hr*4;
    if (t->screen_bytes < screen_frame*nframes) {
        HIP_TRY(hipStreamSynchronize(p->ctx->stream));
        hipFree(t->d_screen); t->d_screen = nullptr; t->screen_bytes = 0;
        HIP_TRY(hipMalloc(&t->d_screen, screen_frame*nframes));
        t->screen_bytes = screen_frame*nframes;
    }
    a.out = t->d_screen; a.out_frame_stride = (long)screen_frame; a.out_components = 4; a.out_dtype = DT_U8;
    if ((rc = launch_render_p(p, a, nframes, p->ctx->stream))) return rc;
    ResolveArgs r;
    r.screen = Tex{t->d_screen, a.wr, a.hr, 4, DT_U8, p->ctx->filter_model == SFX_FILTER_FIXED8 ? FILTER_LINEAR_FIXED8 : FILTER_LINEAR, 0, 0};      // iScreen: linear, repeat(False) (scene.py:192-194)
    r.w = width; r.h = height; r.subsample = subsample; r.out = (uint8_t*)device_out;
    r.screen_frame_stride = (long)screen_frame; r.out_frame_stride = (long)width*height*3; r.top_down = p->ctx->top_down;
    if ((rc = sfl::launch_resolve(p->ctx, r, nframes, p->ctx->stream))) return rc;
    return launch_status();
}
