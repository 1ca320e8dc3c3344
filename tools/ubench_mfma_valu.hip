// Micro-benchmark: do v_mfma_f32_4x4x4_16b_f16 instructions overlap with plain f32 VALU work of the same SIMD on gfx950?
// Per loop iteration: 48 independent v_fma_f32 (8 chains) and M MFMAs (3 accumulators), 8 waves per SIMD, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_valu.hip -o build/variants/ubench_mfma_valu && build/variants/ubench_mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int NFMA8, int NMFMA>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float x = 1.0001f, y = 0.9999f;
    half4 w = {(_Float16)1.0f, (_Float16)0.3f, (_Float16)0.6f, (_Float16)0.18f}, c = {(_Float16)100.0f, (_Float16)-3.0f, (_Float16)7.0f, (_Float16)2.0f};
    float4v r = {0, 0, 0, 0}, g = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int n = 0; n < NFMA8; n++)
            asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                         "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
#pragma unroll
        for (int n = 0; n < NMFMA; n++) {
            if (n % 3 == 0) r = __builtin_amdgcn_mfma_f32_4x4x4f16(w, c, r, 0, 0, 0);
            if (n % 3 == 1) g = __builtin_amdgcn_mfma_f32_4x4x4f16(w, c, g, 0, 0, 0);
            if (n % 3 == 2) b = __builtin_amdgcn_mfma_f32_4x4x4f16(w, c, b, 0, 0, 0);
        }
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + r[0] + g[1] + b[2] + r[3];
}

template <int NFMA8, int NMFMA> void run(const char* name) {
    const int blocks = 256*4, iters = 20000;
    float* out; hipMalloc(&out, blocks*512*sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NFMA8, NMFMA>), dim3(blocks), dim3(512), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NFMA8, NMFMA>), dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD: blocks*8 waves / (256 CUs * 4 SIMDs) = 8; cycles per iteration per wave-on-SIMD at 2.4 GHz nominal
    const double cycles = ms*1e-3*2.4e9/iters/8.0;
    printf("%-28s %8.3f ms   %6.1f cycles per wave-iteration (%d fma + %d mfma)\n", name, ms, cycles, NFMA8*8, NMFMA);
    hipFree(out);
}

int main() {
    run<6, 0>("48 fma");
    run<6, 3>("48 fma + 3 mfma");
    run<6, 6>("48 fma + 6 mfma");
    run<6, 12>("48 fma + 12 mfma");
    run<0, 12>("12 mfma");
    run<2, 12>("16 fma + 12 mfma");
    run<12, 12>("96 fma + 12 mfma");
    return 0;
}
