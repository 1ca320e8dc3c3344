// Micro-benchmark (round 5): the packed-f32 VALU forms (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) on gfx950 — issue cycles per wave64
// instruction per SIMD with vector-pair sources and with a SCALAR pair broadcast through op_sel, and what such a broadcast computes.
// The question behind it: could the strip kernel's per-row fractions (scalar registers) feed two channels' fmas at once without the
// v_mov v, s each of them costs today (profiles/r05_ubench_valu_sgpr.txt: 4.2 cycles)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_pk_f32.hip -o build/ubench_pk_f32 && build/ubench_pk_f32
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHAIN8(INSTR)                                                                                          \
    asm volatile(INSTR("%0") "\n" INSTR("%1") "\n" INSTR("%2") "\n" INSTR("%3") "\n"                          \
                 INSTR("%4") "\n" INSTR("%5") "\n" INSTR("%6") "\n" INSTR("%7") "\n"                          \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y), "s"(sp));
#define R4(X) X X X X

#define I_PK_FMA(r) "v_pk_fma_f32 " r ", %8, %9, " r
#define I_PK_FMA_S(r) "v_pk_fma_f32 " r ", %10, %9, " r " op_sel_hi:[0,1,1]"
#define I_PK_FMA_S_HI(r) "v_pk_fma_f32 " r ", %10, %9, " r " op_sel:[1,0,0] op_sel_hi:[1,1,1]"
#define I_PK_FMA_VB(r) "v_pk_fma_f32 " r ", %8, %9, " r " op_sel_hi:[0,1,1]"
#define I_PK_ADD(r) "v_pk_add_f32 " r ", " r ", %8"
#define I_PK_MUL(r) "v_pk_mul_f32 " r ", " r ", %8"

template <int WHICH>
__global__ __launch_bounds__(512) void k(float* out, int iters, float s0, float s1) {
    const float t = threadIdx.x;
    f2 a0 = {t, t + 1}, a1 = {t + 2, t + 3}, a2 = {t + 4, t + 5}, a3 = {t + 6, t + 7}, a4 = {t + 8, t + 9}, a5 = {t + 10, t + 11}, a6 = {t + 12, t + 13}, a7 = {t + 14, t + 15};
    f2 x = {1.0001f, 0.9998f}, y = {0.9999f, 1.0002f};
    f2 sp = {s0, s1};
    for (int i = 0; i < iters; i++) {
        if constexpr (WHICH == 0) { R4(CHAIN8(I_PK_FMA)) }
        if constexpr (WHICH == 1) { R4(CHAIN8(I_PK_FMA_S)) }
        if constexpr (WHICH == 2) { R4(CHAIN8(I_PK_FMA_VB)) }
        if constexpr (WHICH == 3) { R4(CHAIN8(I_PK_ADD)) }
        if constexpr (WHICH == 4) { R4(CHAIN8(I_PK_MUL)) }
        if constexpr (WHICH == 5) { R4(CHAIN8(I_PK_FMA_S_HI)) }
    }
    const f2 sum = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x*blockDim.x + threadIdx.x] = sum.x + sum.y;
}

// what a scalar-pair source with op_sel computes: acc = {1, 2}, s = {3, 5}, y = {7, 11}
__global__ void semantics(float* out, float s0, float s1) {
    f2 sp = {s0, s1}, y = {7.0f, 11.0f};
    f2 lo = {1.0f, 2.0f}, hi = {1.0f, 2.0f}, plain = {1.0f, 2.0f};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(lo) : "s"(sp), "v"(y));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(hi) : "s"(sp), "v"(y));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(plain) : "s"(sp), "v"(y));
    if (threadIdx.x == 0) { out[0] = lo.x; out[1] = lo.y; out[2] = hi.x; out[3] = hi.y; out[4] = plain.x; out[5] = plain.y; }
}

template <int WHICH> void run(const char* name, float* d_out, int cus, double ghz) {
    const int iters = 4096, blocks = cus*4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(512), 0, 0, d_out, 64, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(512), 0, 0, d_out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms*1e6/((double)iters*32*8);
    printf("%-58s %8.3f ms   %6.3f ns per wave-instruction per SIMD  = %5.2f cycles @ %.2f GHz\n", name, ms, ns, ns*ghz, ghz);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate/1e6;
    printf("%s: %d CUs, %.2f GHz max\n", prop.name, cus, ghz);
    float* d_out; hipMalloc(&d_out, sizeof(float)*cus*4*512);
    run<0>("v_pk_fma_f32 v[2], v[2], v[2], v[2]", d_out, cus, ghz);
    run<1>("v_pk_fma_f32 v[2], s[2] (lo broadcast), v[2], v[2]", d_out, cus, ghz);
    run<5>("v_pk_fma_f32 v[2], s[2] (hi broadcast), v[2], v[2]", d_out, cus, ghz);
    run<2>("v_pk_fma_f32 v[2], v[2] (lo broadcast), v[2], v[2]", d_out, cus, ghz);
    run<3>("v_pk_add_f32 v[2], v[2], v[2]", d_out, cus, ghz);
    run<4>("v_pk_mul_f32 v[2], v[2], v[2]", d_out, cus, ghz);
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, d_out, 3.0f, 5.0f);
    float h[6]; hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    printf("acc {1, 2} + s {3, 5} * y {7, 11}: lo broadcast -> {%g, %g} (22, 35 if s.lo feeds both), hi broadcast -> {%g, %g} (36, 57 if s.hi feeds both), no op_sel -> {%g, %g} (22, 57 if halves pair up)\n",
           h[0], h[1], h[2], h[3], h[4], h[5]);
    return 0;
}
