// Does v_cvt_pk_u8_f32 do what glsl.hpp unorm8() does after the multiplication — clamp to [0, 255], round half to even, NaN -> 0 —
// for every float? (It would replace max + min + rint + cvt + shift/or per channel.)  hipcc --offload-arch=gfx950 -O2 tools/check_cvt_pk_u8.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
__global__ void k(unsigned long long* bad, unsigned* first) {
    const unsigned long long n = 1ull << 32;
    for (unsigned long long i = blockIdx.x*(unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x*blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        float c = x;
        c = (c > 0.0f) ? c : 0.0f;                 // glsl.hpp unorm8 on the already scaled value: clamp to [0, 255]
        c = (c < 255.0f) ? c : 255.0f;
        const unsigned want = (unsigned)rintf(c);
        unsigned got;
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(got) : "v"(x));
        if ((got & 255u) != want) { atomicAdd(bad, 1ull); atomicMin(first, (unsigned)i); }
    }
}
int main() {
    unsigned long long* bad; unsigned* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4);
    hipMemset(bad, 0, 8); hipMemset(first, 0xff, 4);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, bad, first);
    unsigned long long b; unsigned f;
    hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
    float x; memcpy(&x, &f, 4);
    printf("v_cvt_pk_u8_f32 vs clamp+rint over all 2^32 floats: %llu mismatches%s\n", b, b ? "" : " (identical)");
    if (b) printf("  first mismatch: x = %g (bits %08x)\n", x, f);
    return 0;
}
