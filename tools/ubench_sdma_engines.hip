// Device-to-host throughput of every SDMA engine of the GPU, alone and beside a kernel that fills the chip (round 4: why a HIP copy
// stream sometimes reads 4K frames out at 20 GB/s instead of 55). HSA's copy-on-engine API, 60 frames of 24.9 MB per measurement.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_sdma_engines.hip -o build/ubench_sdma_engines -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define HK(x) do { hsa_status_t e = (x); if (e != HSA_STATUS_SUCCESS) { const char* m = ""; hsa_status_string(e, &m); printf("%s: %s\n", #x, m); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_busy(float* out, int iters) {
    float a = threadIdx.x*1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; i++) { a = a*b + 0.5f; b = b*0.9999f + 1e-4f; }
    if (a == 12345.0f) out[0] = a + b;
}

static std::vector<hsa_agent_t> gpus, cpus;
static hsa_status_t on_agent(hsa_agent_t agent, void*) {
    hsa_device_type_t type; hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type);
    (type == HSA_DEVICE_TYPE_GPU ? gpus : cpus).push_back(agent);
    return HSA_STATUS_SUCCESS;
}

int main() {
    const size_t frame = 3840ull*2160*3; const int frames = 60;
    uint8_t *dev, *host; float* scratch;
    CK(hipMalloc(&dev, frame*8)); CK(hipMemset(dev, 1, frame*8)); CK(hipHostMalloc(&host, frame*8, hipHostMallocDefault)); CK(hipMalloc(&scratch, 4096));
    for (size_t i = 0; i < frame*8; i += 4096) host[i] = 0;
    HK(hsa_init());
    HK(hsa_iterate_agents(on_agent, nullptr));
    hsa_agent_t gpu = gpus[0], cpu = cpus[0];
    uint32_t free_mask = 0, preferred = 0;
    HK(hsa_amd_memory_copy_engine_status(cpu, gpu, &free_mask));
    hsa_amd_memory_get_preferred_copy_engine(cpu, gpu, &preferred);
    printf("device-to-host: free engine mask 0x%x, preferred 0x%x; %zu gpu agent(s), %zu cpu agent(s)\n", free_mask, preferred, gpus.size(), cpus.size());
    hipStream_t busy; CK(hipStreamCreateWithFlags(&busy, hipStreamNonBlocking));
    for (int beside = 0; beside < 2; beside++) {
        printf("## %s\n", beside ? "beside a kernel on every CU" : "idle GPU");
        for (int engine = 0; engine < 16; engine++) {
            if (!(free_mask >> engine & 1)) continue;
            hsa_signal_t done; HK(hsa_signal_create(1, 0, nullptr, &done));
            if (beside) hipLaunchKernelGGL(k_busy, dim3(256*8), dim3(256), 0, busy, scratch, 40000000);     // ≈ a few hundred ms
            const double t0 = now();
            hsa_status_t status = HSA_STATUS_SUCCESS;
            for (int f = 0; f < frames && status == HSA_STATUS_SUCCESS; f++) {
                hsa_signal_store_relaxed(done, 1);
                status = hsa_amd_memory_async_copy_on_engine(host + (size_t)(f % 8)*frame, cpu, dev + (size_t)(f % 8)*frame, gpu, frame, 0, nullptr, done, (hsa_amd_sdma_engine_id_t)(1u << engine), false);
                if (status == HSA_STATUS_SUCCESS) hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
            }
            const double dt = now() - t0;
            const char* message = ""; hsa_status_string(status, &message);
            if (status == HSA_STATUS_SUCCESS) printf("engine %2d (0x%04x): %6.2f GB/s = %6.1f frames/s\n", engine, 1u << engine, frame*(double)frames/dt/1e9, frames/dt);
            else printf("engine %2d (0x%04x): %s\n", engine, 1u << engine, message);
            CK(hipStreamSynchronize(busy));
            hsa_signal_destroy(done);
        }
    }
    // two engines at once (what two HIP copy streams do)
    for (int pair = 0; pair < 4; pair++) {
        const int a = pair == 0 ? 0 : (pair == 1 ? 0 : (pair == 2 ? 2 : 4)), b = pair == 0 ? 1 : (pair == 1 ? 3 : (pair == 2 ? 3 : 5));
        if (!(free_mask >> a & 1) || !(free_mask >> b & 1)) continue;
        hsa_signal_t done[2]; for (auto& d : done) HK(hsa_signal_create(0, 0, nullptr, &d));
        const double t0 = now();
        for (int f = 0; f < frames; f++) {
            hsa_signal_t d = done[f & 1];
            hsa_signal_wait_scacquire(d, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
            hsa_signal_store_relaxed(d, 1);
            HK(hsa_amd_memory_async_copy_on_engine(host + (size_t)(f % 8)*frame, cpu, dev + (size_t)(f % 8)*frame, gpu, frame, 0, nullptr, d, (hsa_amd_sdma_engine_id_t)(1u << ((f & 1) ? b : a)), false));
        }
        for (auto& d : done) hsa_signal_wait_scacquire(d, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
        const double dt = now() - t0;
        printf("engines %d + %d alternating, two in flight: %6.2f GB/s = %6.1f frames/s\n", a, b, frame*(double)frames/dt/1e9, frames/dt);
        for (auto& d : done) hsa_signal_destroy(d);
    }
    return 0;
}
