// Micro-benchmark, second set (round 5's ISA census): issue cost of the VALU forms the strip kernel's "other" 16 % is made of — moves from
// scalar registers, selects on a scalar-pair mask, compares into a scalar pair, adds with a scalar operand, readfirstlane, med3 — in cycles
// per wave64 instruction per SIMD on gfx950. 8 independent chains per wave, 8 waves per SIMD, every CU busy (as tools/ubench_valu.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu_sgpr.hip -o build/ubench_valu_sgpr && build/ubench_valu_sgpr
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(INSTR)                                                                                          \
    asm volatile(INSTR("%0") "\n" INSTR("%1") "\n" INSTR("%2") "\n" INSTR("%3") "\n"                          \
                 INSTR("%4") "\n" INSTR("%5") "\n" INSTR("%6") "\n" INSTR("%7") "\n"                          \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y), "s"(sx), "s"(mask) : "vcc");
#define R4(X) X X X X

#define I_FMA(r) "v_fma_f32 " r ", %8, %9, " r
#define I_MOV_V(r) "v_mov_b32 " r ", %8"
#define I_MOV_S(r) "v_mov_b32 " r ", %10"
#define I_ADD_U32_S(r) "v_add_u32 " r ", %10, " r
#define I_ADD_U32_V(r) "v_add_u32 " r ", %8, " r
#define I_MUL_S(r) "v_mul_f32 " r ", %10, " r
#define I_FMAC_S(r) "v_fmac_f32 " r ", %10, %9"
#define I_CND_SMASK(r) "v_cndmask_b32 " r ", %8, " r ", %11"
#define I_CND_VCC(r) "v_cndmask_b32 " r ", %8, " r ", vcc"
#define I_CMP_VCC(r) "v_cmp_gt_f32 vcc, %8, " r
#define I_CMP_SPAIR(r) "v_cmp_gt_f32 s[20:21], %8, " r
#define I_MED3(r) "v_med3_f32 " r ", " r ", 0, 1.0"
#define I_READFIRST(r) "v_readfirstlane_b32 s22, " r
#define I_CVT_PK_U8(r) "v_cvt_pk_u8_f32 " r ", %8, 1, " r
#define I_SUB_SDWA(r) "v_sub_u32_sdwa " r ", %8, " r " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1"

template <int WHICH>
__global__ __launch_bounds__(512) void k(float* out, int iters, float sxf, unsigned long long m) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float x = 1.0001f, y = 0.9999f;
    float sx = sxf;
    unsigned long long mask = m;
    for (int i = 0; i < iters; i++) {
        if constexpr (WHICH == 0) { R4(CHAIN8(I_FMA)) }
        if constexpr (WHICH == 1) { R4(CHAIN8(I_MOV_V)) }
        if constexpr (WHICH == 2) { R4(CHAIN8(I_MOV_S)) }
        if constexpr (WHICH == 3) { R4(CHAIN8(I_ADD_U32_V)) }
        if constexpr (WHICH == 4) { R4(CHAIN8(I_ADD_U32_S)) }
        if constexpr (WHICH == 5) { R4(CHAIN8(I_MUL_S)) }
        if constexpr (WHICH == 6) { R4(CHAIN8(I_FMAC_S)) }
        if constexpr (WHICH == 7) { R4(CHAIN8(I_CND_SMASK)) }
        if constexpr (WHICH == 8) { R4(CHAIN8(I_CND_VCC)) }
        if constexpr (WHICH == 9) { R4(CHAIN8(I_CMP_VCC)) }
        if constexpr (WHICH == 10) { asm volatile("" ::: "s20", "s21"); R4(CHAIN8(I_CMP_SPAIR)) }
        if constexpr (WHICH == 11) { R4(CHAIN8(I_MED3)) }
        if constexpr (WHICH == 12) { asm volatile("" ::: "s22"); R4(CHAIN8(I_READFIRST)) }
        if constexpr (WHICH == 13) { R4(CHAIN8(I_CVT_PK_U8)) }
        if constexpr (WHICH == 14) { R4(CHAIN8(I_SUB_SDWA)) }
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int WHICH> void run(const char* name, float* d_out, int cus, double ghz) {
    const int iters = 4096, blocks = cus*4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(512), 0, 0, d_out, 64, 1.0f, 0x5555aaaa5555aaaaull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(512), 0, 0, d_out, iters, 1.0f, 0x5555aaaa5555aaaaull);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms*1e6/((double)iters*32*8);
    printf("%-34s %8.3f ms   %6.3f ns per wave-instruction per SIMD  = %5.2f cycles @ %.2f GHz\n", name, ms, ns, ns*ghz, ghz);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate/1e6;
    printf("%s: %d CUs, %.2f GHz max\n", prop.name, cus, ghz);
    float* d_out; hipMalloc(&d_out, sizeof(float)*cus*4*512);
    run<0>("v_fma_f32 (reference)", d_out, cus, ghz);
    run<1>("v_mov_b32 v, v", d_out, cus, ghz);
    run<2>("v_mov_b32 v, s", d_out, cus, ghz);
    run<3>("v_add_u32 v, v, v", d_out, cus, ghz);
    run<4>("v_add_u32 v, s, v", d_out, cus, ghz);
    run<5>("v_mul_f32 v, s, v", d_out, cus, ghz);
    run<6>("v_fmac_f32 v, s, v", d_out, cus, ghz);
    run<7>("v_cndmask_b32 v, v, v, s[pair]", d_out, cus, ghz);
    run<8>("v_cndmask_b32 v, v, v, vcc", d_out, cus, ghz);
    run<9>("v_cmp_gt_f32 vcc, v, v", d_out, cus, ghz);
    run<10>("v_cmp_gt_f32 s[pair], v, v", d_out, cus, ghz);
    run<11>("v_med3_f32 v, v, 0, 1.0", d_out, cus, ghz);
    run<12>("v_readfirstlane_b32 s, v", d_out, cus, ghz);
    run<13>("v_cvt_pk_u8_f32", d_out, cus, ghz);
    run<14>("v_sub_u32_sdwa (byte selects)", d_out, cus, ghz);
    return 0;
}
