// capi_readout.hip — the C-ABI's read-out and peer-copy half (include/shaderflow_hip.h: sfx_ring_*, sfx_shm_*, sfx_peer_*,
// sfx_rgb_to_yuv420, sfx_device_copy, sfx_ctx_copy_streams): how finished frames leave device memory. Host-side state and three
// small kernels of its own; nothing here launches a fragment. (One of the library's translation units: capi.hip keeps contexts,
// textures, programs and the render dispatch, capi_audio.hip the audio plans and the tape.)

#include "host_state.hpp"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <unistd.h>

using namespace sf;

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

// a frame from one device buffer to another, in order with the context's stream
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

// the context is going away (sfx_ctx_destroy): its peer copier first (its lanes may still sit on the copy streams), then the streams and
// the engine records
void readout_release(Context* c) {
    peer_stop(c);
    for (hipStream_t& stream : c->copy_streams) if (stream) { hipStreamSynchronize(stream); hipStreamDestroy(stream); stream = nullptr; }
    delete c->engines; c->engines = nullptr;
    delete c->peer_engines; c->peer_engines = nullptr;
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
int ring_slot_count(sfx_handle h) {
    Ring* r = get<Ring>(h, MAGIC_RING);
    return r ? r->slots : -1;
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
