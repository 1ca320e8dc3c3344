// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the blur loop can be
// built from, on gfx950. 8 independent chains per wave, 8 waves per SIMD, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o build/ubench_valu && build/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHAIN8(INSTR)                                                                                          \
    asm volatile(INSTR("%0") "\n" INSTR("%1") "\n" INSTR("%2") "\n" INSTR("%3") "\n"                          \
                 INSTR("%4") "\n" INSTR("%5") "\n" INSTR("%6") "\n" INSTR("%7") "\n"                          \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y), "s"(sx));

#define I_FMA(r) "v_fma_f32 " r ", %8, %9, " r
#define I_FMA_MIX(r) "v_fma_mix_f32 " r ", %8, %9, " r " op_sel_hi:[0,1,0]"
#define I_FMA_MIX_HI(r) "v_fma_mix_f32 " r ", %8, %9, " r " op_sel:[0,1,0] op_sel_hi:[0,1,0]"
#define I_CVT_F16(r) "v_cvt_f32_f16 " r ", %9"
#define I_DOT2_F16(r) "v_dot2_f32_f16 " r ", %8, %9, " r
#define I_DOT2C_F16(r) "v_dot2c_f32_f16 " r ", %8, %9"
#define I_PK_FMA_F32(r) "v_fma_f32 " r ", %8, %9, " r
#define I_PK_FMA_F16(r) "v_pk_fma_f16 " r ", %8, %9, " r
#define I_FRACT(r) "v_fract_f32 " r ", " r
#define I_CVT_I32(r) "v_cvt_i32_f32 " r ", " r
#define I_MAD_U24(r) "v_mad_u32_u24 " r ", %8, %9, " r
#define I_LSHL_ADD(r) "v_lshl_add_u32 " r ", %8, 5, " r
#define I_DOT4_U8(r) "v_dot4_u32_u8 " r ", %8, %9, " r
#define I_DOT2_U16(r) "v_dot2_u32_u16 " r ", %8, %9, " r
#define I_MUL(r) "v_mul_f32 " r ", %8, " r
#define I_ADD(r) "v_add_f32 " r ", %8, " r
#define I_FMA_S(r) "v_fma_f32 " r ", %10, %9, " r
#define I_PERM(r) "v_perm_b32 " r ", %8, %9, " r
#define I_CVT_UBYTE(r) "v_cvt_f32_ubyte1 " r ", %9"
#define I_FMA_LIT(r) "v_fma_f32 " r ", 0x3f9d70a4, %9, " r
#define I_FMA_INL(r) "v_fma_f32 " r ", 2.0, %9, " r
#define I_ADD_INL(r) "v_add_f32 " r ", 1.0, " r
#define I_MUL_LIT(r) "v_mul_f32 " r ", 0x45800000, " r
#define I_SUB(r) "v_sub_f32 " r ", %8, " r
#define I_FLOOR(r) "v_floor_f32 " r ", " r
#define I_ADD_U32(r) "v_add_u32 " r ", %8, " r
#define I_AND(r) "v_and_b32 " r ", %8, " r
#define I_LSHL(r) "v_lshlrev_b32 " r ", 5, " r
#define I_CVT_PKRTZ(r) "v_cvt_pkrtz_f16_f32 " r ", %8, " r
#define I_MAX(r) "v_max_f32 " r ", %8, " r
#define I_CNDMASK(r) "v_cndmask_b32 " r ", %8, " r ", vcc"
#define I_RCP(r) "v_rcp_f32 " r ", " r
#define I_SQRT(r) "v_sqrt_f32 " r ", " r
#define I_FMAC(r) "v_fmac_f32 " r ", %8, %9"
#define I_FMAAK(r) "v_fmaak_f32 " r ", %8, " r ", 0x3f9d70a4"
#define I_MOV(r) "v_mov_b32 " r ", %8"
#define I_PK_FMA32(r) "v_pk_fma_f32 %0, %8, %9, %0 \n v_pk_fma_f32 %1, %8, %9, %1"
#define I_MAD_MIX_LEGACY(r) "v_fma_mix_f32 " r ", %8, %9, %8 op_sel_hi:[0,1,0]"

template <int WHICH>
__global__ __launch_bounds__(512) void k(float* out, int iters, float sxf) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float x = 1.0001f, y = 0.9999f;
    float sx = sxf;
    for (int i = 0; i < iters; i++) {
        if constexpr (WHICH == 0) { CHAIN8(I_FMA) CHAIN8(I_FMA) CHAIN8(I_FMA) CHAIN8(I_FMA) }
        if constexpr (WHICH == 1) { CHAIN8(I_FMA_MIX) CHAIN8(I_FMA_MIX) CHAIN8(I_FMA_MIX) CHAIN8(I_FMA_MIX) }
        if constexpr (WHICH == 2) { CHAIN8(I_FMA_MIX_HI) CHAIN8(I_FMA_MIX_HI) CHAIN8(I_FMA_MIX_HI) CHAIN8(I_FMA_MIX_HI) }
        if constexpr (WHICH == 3) { CHAIN8(I_CVT_F16) CHAIN8(I_CVT_F16) CHAIN8(I_CVT_F16) CHAIN8(I_CVT_F16) }
        if constexpr (WHICH == 4) { CHAIN8(I_DOT2_F16) CHAIN8(I_DOT2_F16) CHAIN8(I_DOT2_F16) CHAIN8(I_DOT2_F16) }
        if constexpr (WHICH == 5) { CHAIN8(I_PK_FMA_F16) CHAIN8(I_PK_FMA_F16) CHAIN8(I_PK_FMA_F16) CHAIN8(I_PK_FMA_F16) }
        if constexpr (WHICH == 6) { CHAIN8(I_FRACT) CHAIN8(I_FRACT) CHAIN8(I_FRACT) CHAIN8(I_FRACT) }
        if constexpr (WHICH == 7) { CHAIN8(I_CVT_I32) CHAIN8(I_CVT_I32) CHAIN8(I_CVT_I32) CHAIN8(I_CVT_I32) }
        if constexpr (WHICH == 8) { CHAIN8(I_MAD_U24) CHAIN8(I_MAD_U24) CHAIN8(I_MAD_U24) CHAIN8(I_MAD_U24) }
        if constexpr (WHICH == 9) { CHAIN8(I_LSHL_ADD) CHAIN8(I_LSHL_ADD) CHAIN8(I_LSHL_ADD) CHAIN8(I_LSHL_ADD) }
        if constexpr (WHICH == 10) { CHAIN8(I_DOT4_U8) CHAIN8(I_DOT4_U8) CHAIN8(I_DOT4_U8) CHAIN8(I_DOT4_U8) }
        if constexpr (WHICH == 11) { CHAIN8(I_DOT2_U16) CHAIN8(I_DOT2_U16) CHAIN8(I_DOT2_U16) CHAIN8(I_DOT2_U16) }
        if constexpr (WHICH == 12) { CHAIN8(I_MUL) CHAIN8(I_MUL) CHAIN8(I_MUL) CHAIN8(I_MUL) }
        if constexpr (WHICH == 13) { CHAIN8(I_FMA_S) CHAIN8(I_FMA_S) CHAIN8(I_FMA_S) CHAIN8(I_FMA_S) }
        if constexpr (WHICH == 14) { CHAIN8(I_PERM) CHAIN8(I_PERM) CHAIN8(I_PERM) CHAIN8(I_PERM) }
        if constexpr (WHICH == 15) { CHAIN8(I_CVT_UBYTE) CHAIN8(I_CVT_UBYTE) CHAIN8(I_CVT_UBYTE) CHAIN8(I_CVT_UBYTE) }
        if constexpr (WHICH == 16) { CHAIN8(I_DOT2C_F16) CHAIN8(I_DOT2C_F16) CHAIN8(I_DOT2C_F16) CHAIN8(I_DOT2C_F16) }
        if constexpr (WHICH == 21) { CHAIN8(I_FMA_INL) CHAIN8(I_FMA_INL) CHAIN8(I_FMA_INL) CHAIN8(I_FMA_INL) }
        if constexpr (WHICH == 22) { CHAIN8(I_ADD_INL) CHAIN8(I_ADD_INL) CHAIN8(I_ADD_INL) CHAIN8(I_ADD_INL) }
        if constexpr (WHICH == 23) { CHAIN8(I_MUL_LIT) CHAIN8(I_MUL_LIT) CHAIN8(I_MUL_LIT) CHAIN8(I_MUL_LIT) }
        if constexpr (WHICH == 24) { CHAIN8(I_SUB) CHAIN8(I_SUB) CHAIN8(I_SUB) CHAIN8(I_SUB) }
        if constexpr (WHICH == 25) { CHAIN8(I_FLOOR) CHAIN8(I_FLOOR) CHAIN8(I_FLOOR) CHAIN8(I_FLOOR) }
        if constexpr (WHICH == 26) { CHAIN8(I_ADD_U32) CHAIN8(I_ADD_U32) CHAIN8(I_ADD_U32) CHAIN8(I_ADD_U32) }
        if constexpr (WHICH == 27) { CHAIN8(I_AND) CHAIN8(I_AND) CHAIN8(I_AND) CHAIN8(I_AND) }
        if constexpr (WHICH == 28) { CHAIN8(I_LSHL) CHAIN8(I_LSHL) CHAIN8(I_LSHL) CHAIN8(I_LSHL) }
        if constexpr (WHICH == 29) { CHAIN8(I_CVT_PKRTZ) CHAIN8(I_CVT_PKRTZ) CHAIN8(I_CVT_PKRTZ) CHAIN8(I_CVT_PKRTZ) }
        if constexpr (WHICH == 30) { CHAIN8(I_MAX) CHAIN8(I_MAX) CHAIN8(I_MAX) CHAIN8(I_MAX) }
        if constexpr (WHICH == 31) { CHAIN8(I_CNDMASK) CHAIN8(I_CNDMASK) CHAIN8(I_CNDMASK) CHAIN8(I_CNDMASK) }
        if constexpr (WHICH == 32) { CHAIN8(I_RCP) CHAIN8(I_RCP) CHAIN8(I_RCP) CHAIN8(I_RCP) }
        if constexpr (WHICH == 33) { CHAIN8(I_SQRT) CHAIN8(I_SQRT) CHAIN8(I_SQRT) CHAIN8(I_SQRT) }
        if constexpr (WHICH == 34) { CHAIN8(I_FMAC) CHAIN8(I_FMAC) CHAIN8(I_FMAC) CHAIN8(I_FMAC) }
        if constexpr (WHICH == 35) { CHAIN8(I_FMAAK) CHAIN8(I_FMAAK) CHAIN8(I_FMAAK) CHAIN8(I_FMAAK) }
        if constexpr (WHICH == 36) { CHAIN8(I_MOV) CHAIN8(I_MOV) CHAIN8(I_MOV) CHAIN8(I_MOV) }
        if constexpr (WHICH == 37) { CHAIN8(I_ADD) CHAIN8(I_ADD) CHAIN8(I_ADD) CHAIN8(I_ADD) }
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ __launch_bounds__(512) void k_pk(double* out, int iters) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double x = 1.0001, y = 0.9999;
    for (int i = 0; i < iters; i++) {
#define PK8(OP) asm volatile(OP " %0, %8, %9, %0\n" OP " %1, %8, %9, %1\n" OP " %2, %8, %9, %2\n" OP " %3, %8, %9, %3\n" OP " %4, %8, %9, %4\n" OP " %5, %8, %9, %5\n" OP " %6, %8, %9, %6\n" OP " %7, %8, %9, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        PK8("v_pk_fma_f32") PK8("v_pk_fma_f32") PK8("v_pk_fma_f32") PK8("v_pk_fma_f32")
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ __launch_bounds__(512) void k_pkadd(double* out, int iters) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double x = 1.0001;
    for (int i = 0; i < iters; i++) {
#define PA8(OP) asm volatile(OP " %0, %8, %0\n" OP " %1, %8, %1\n" OP " %2, %8, %2\n" OP " %3, %8, %3\n" OP " %4, %8, %4\n" OP " %5, %8, %5\n" OP " %6, %8, %6\n" OP " %7, %8, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
        PA8("v_pk_add_f32") PA8("v_pk_add_f32") PA8("v_pk_add_f32") PA8("v_pk_add_f32")
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int WHICH> double run(const char* name, float* d_out, int cus, double clock_ghz) {
    const int iters = 4096, blocks = cus*4;                  // 4 blocks x 8 waves = 32 waves per CU = 8 per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(512), 0, 0, d_out, 64, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(512), 0, 0, d_out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters*32*8;        // 32 instr per iteration per wave, 8 waves per SIMD
    const double ns_per_instr = ms*1e6/instr_per_simd;
    printf("%-28s %8.3f ms   %6.3f ns per wave-instruction per SIMD  = %5.2f cycles @ %.2f GHz\n", name, ms, ns_per_instr, ns_per_instr*clock_ghz, clock_ghz);
    return ns_per_instr;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate/1e6;
    printf("%s: %d CUs, %.2f GHz max\n", prop.name, cus, ghz);
    float* d_out; hipMalloc(&d_out, sizeof(float)*cus*4*512);
    {
        double* d2; hipMalloc(&d2, sizeof(double)*cus*4*512);
        for (int which = 0; which < 2; which++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            if (which == 0) hipLaunchKernelGGL(k_pk, dim3(cus*4), dim3(512), 0, 0, d2, 64); else hipLaunchKernelGGL(k_pkadd, dim3(cus*4), dim3(512), 0, 0, d2, 64);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_pk, dim3(cus*4), dim3(512), 0, 0, d2, 4096); else hipLaunchKernelGGL(k_pkadd, dim3(cus*4), dim3(512), 0, 0, d2, 4096);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double ns = ms*1e6/(4096.0*32*8);
            printf("%-28s %8.3f ms   %6.3f ns per wave-instruction per SIMD  = %5.2f cycles @ %.2f GHz (2 lanes-ops per lane)\n", which == 0 ? "v_pk_fma_f32" : "v_pk_add_f32", ms, ns, ns*ghz, ghz);
        }
    }
    run<0>("v_fma_f32", d_out, cus, ghz);
    run<13>("v_fma_f32 (sgpr operand)", d_out, cus, ghz);
    run<12>("v_mul_f32", d_out, cus, ghz);
    run<1>("v_fma_mix_f32 (f16 lo)", d_out, cus, ghz);
    run<2>("v_fma_mix_f32 (f16 hi)", d_out, cus, ghz);
    run<3>("v_cvt_f32_f16", d_out, cus, ghz);
    run<15>("v_cvt_f32_ubyte1", d_out, cus, ghz);
    run<4>("v_dot2_f32_f16", d_out, cus, ghz);
    run<16>("v_dot2c_f32_f16", d_out, cus, ghz);
    run<5>("v_pk_fma_f16", d_out, cus, ghz);
    run<6>("v_fract_f32", d_out, cus, ghz);
    run<7>("v_cvt_i32_f32", d_out, cus, ghz);
    run<8>("v_mad_u32_u24", d_out, cus, ghz);
    run<9>("v_lshl_add_u32", d_out, cus, ghz);
    run<10>("v_dot4_u32_u8", d_out, cus, ghz);
    run<11>("v_dot2_u32_u16", d_out, cus, ghz);
    run<14>("v_perm_b32", d_out, cus, ghz);
    run<21>("v_fma_f32 (inline 2.0)", d_out, cus, ghz);
    run<22>("v_add_f32 (inline 1.0)", d_out, cus, ghz);
    run<37>("v_add_f32", d_out, cus, ghz);
    run<23>("v_mul_f32 (literal)", d_out, cus, ghz);
    run<24>("v_sub_f32", d_out, cus, ghz);
    run<25>("v_floor_f32", d_out, cus, ghz);
    run<26>("v_add_u32", d_out, cus, ghz);
    run<27>("v_and_b32", d_out, cus, ghz);
    run<28>("v_lshlrev_b32", d_out, cus, ghz);
    run<29>("v_cvt_pkrtz_f16_f32", d_out, cus, ghz);
    run<30>("v_max_f32", d_out, cus, ghz);
    run<31>("v_cndmask_b32", d_out, cus, ghz);
    run<32>("v_rcp_f32", d_out, cus, ghz);
    run<33>("v_sqrt_f32", d_out, cus, ghz);
    run<34>("v_fmac_f32", d_out, cus, ghz);
    run<35>("v_fmaak_f32", d_out, cus, ghz);
    run<36>("v_mov_b32", d_out, cus, ghz);
    return 0;
}
