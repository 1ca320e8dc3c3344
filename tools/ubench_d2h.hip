// Device-to-host read-out ceiling for finished frames (24.9 MB of RGB8 each at 4K): hipMemcpyAsync into pinned memory on one and
// two streams, chunked, and a copy kernel storing straight into mapped host memory. Prints GB/s and frames/s.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_d2h.hip -o build/ubench_d2h
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x)
        dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t frame = 3840ull*2160*3;
    const int frames = 60, slots = 8, rounds = 4;
    uint8_t* dev; CK(hipMalloc(&dev, frame*frames)); CK(hipMemset(dev, 1, frame*frames));
    for (int flags_i = 0; flags_i < 2; flags_i++) {
        const unsigned flags = flags_i == 0 ? hipHostMallocDefault : (hipHostMallocMapped | hipHostMallocNonCoherent);
        uint8_t* host; CK(hipHostMalloc(&host, frame*slots, flags));
        for (size_t i = 0; i < frame*slots; i += 4096) host[i] = 0;
        printf("## pinned ring of %d slots, hipHostMalloc flags %s\n", slots, flags_i == 0 ? "default" : "mapped|noncoherent");
        for (int nstreams = 1; nstreams <= 4; nstreams *= 2) {
            for (int pieces = 1; pieces <= 4; pieces *= 4) {
                std::vector<hipStream_t> st(nstreams);
                for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
                CK(hipDeviceSynchronize());
                const double t0 = now();
                for (int r = 0; r < rounds; r++)
                    for (int f = 0; f < frames; f++)
                        for (int p = 0; p < pieces; p++) {
                            const size_t off = frame/pieces*p, len = (p == pieces - 1) ? frame - off : frame/pieces;
                            CK(hipMemcpyAsync(host + (size_t)(f % slots)*frame + off, dev + (size_t)f*frame + off, len, hipMemcpyDeviceToHost, st[(f*pieces + p) % nstreams]));
                        }
                CK(hipDeviceSynchronize());
                const double dt = now() - t0, bytes = (double)frame*frames*rounds;
                printf("hipMemcpyAsync  %d stream(s) %d piece(s) per frame   %7.2f GB/s   %7.1f frames/s\n", nstreams, pieces, bytes/dt/1e9, frames*rounds/dt);
                for (auto& s : st) CK(hipStreamDestroy(s));
            }
        }
        uint8_t* mapped = nullptr;
        if (hipHostGetDevicePointer((void**)&mapped, host, 0) == hipSuccess) {
            for (int blocks = 64; blocks <= 4096; blocks *= 4) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                for (int r = 0; r < rounds; r++)
                    for (int f = 0; f < frames; f++)
                        hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, (const uint4*)(dev + (size_t)f*frame), (uint4*)(mapped + (size_t)(f % slots)*frame), frame/16);
                CK(hipDeviceSynchronize());
                const double dt = now() - t0, bytes = (double)frame*frames*rounds;
                printf("copy kernel into mapped host memory, %4d blocks        %7.2f GB/s   %7.1f frames/s\n", blocks, bytes/dt/1e9, frames*rounds/dt);
            }
        }
        CK(hipHostFree(host));
    }
    return 0;
}
