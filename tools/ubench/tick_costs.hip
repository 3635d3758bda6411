// What rides behind an MFMA on a lone wave (diagnostic, not part of the product; cited in DESIGN.md §6.0).
// One wave per SIMD runs the register-resident-basis kernel's tick shape: v_mfma_f32_16x16x32_f16 with A and C/D in the
// accumulation registers and B architectural, four accumulators round robin, and behind every MFMA N instructions of one
// kind on independent registers.  Prints shader clocks per tick.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/tick_costs tools/ubench/tick_costs.hip && gpurun -- ./tools/ubench/tick_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define MFMA(m) "v_mfma_f32_16x16x32_f16 %" #m ", %4, %5, %" #m "\n"
// operands: %0..%3 accumulators (a), %4 A (a), %5 B (v), %6..%13 eight scratch registers (v), %14 constant (v), %15 packed constant (v),
// %16..%19 packed scratch (v), %20 LDS address (v)
#define KERNEL(NAME, X0, X1, X2, X3)                                                                                       \
    __global__ void __launch_bounds__(256) NAME(int iters, unsigned long long *cyc, float *sink)                           \
    {                                                                                                                      \
        __shared__ float lds[4096];                                                                                        \
        const int lane = threadIdx.x & 63;                                                                                 \
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;                                                          \
        __syncthreads();                                                                                                   \
        floatx4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;                                                               \
        uint32x4 A = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, B = A;                                            \
        float r0 = lane, r1 = lane + 1, r2 = lane + 2, r3 = lane + 3, r4 = lane + 4, r5 = lane + 5, r6 = lane + 6, r7 = lane + 7; \
        f2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7};                                                      \
        float c = 1.0000001f; f2 cc = {c, c};                                                                              \
        unsigned addr = (unsigned)(uintptr_t)(lds + 4 * (threadIdx.x & 255));                                              \
        uint32x4 qq = A; unsigned voff = 16 * threadIdx.x;                                                                 \
        asm volatile("s_mov_b32 s24, %0\ns_mov_b32 s25, %1\ns_mov_b32 s26, 0x10000\ns_mov_b32 s27, 0x00020000" : : "s"((unsigned)(uintptr_t)sink), "s"((unsigned)((uintptr_t)sink >> 32) & 0xffffu) : "s24", "s25", "s26", "s27"); \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                        \
        for (int it = 0; it < iters; it++) {                                                                               \
            asm volatile(MFMA(0) X0 MFMA(1) X1 MFMA(2) X2 MFMA(3) X3 MFMA(0) X0 MFMA(1) X1 MFMA(2) X2 MFMA(3) X3             \
                         : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3)                                                          \
                         : "a"(A), "v"(B), "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(c), "v"(cc),  \
                           "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(addr), "v"(qq), "v"(voff)                                                 \
                         : "memory", "s20", "s21", "s24", "s25", "s26", "s27");                                                                                      \
        }                                                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                        \
        if (lane == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;                                                   \
        float s = c0[0] + c1[0] + c2[0] + c3[0];                                                                           \
        if (s == 12345.678f) sink[0] = s + r0 + p0.x;                                                                      \
    }
// (the scratch registers are inputs only as far as the compiler knows: the statement overwrites them, nothing reads them after)

KERNEL(k_none, "", "", "", "")
KERNEL(k_mul1, "v_mul_f32 %6, %6, %14\n", "v_mul_f32 %7, %7, %14\n", "v_mul_f32 %8, %8, %14\n", "v_mul_f32 %9, %9, %14\n")
KERNEL(k_mul2, "v_mul_f32 %6, %6, %14\nv_mul_f32 %10, %10, %14\n", "v_mul_f32 %7, %7, %14\nv_mul_f32 %11, %11, %14\n", "v_mul_f32 %8, %8, %14\nv_mul_f32 %12, %12, %14\n", "v_mul_f32 %9, %9, %14\nv_mul_f32 %13, %13, %14\n")
KERNEL(k_mul3, "v_mul_f32 %6, %6, %14\nv_mul_f32 %10, %10, %14\nv_mul_f32 %7, %7, %14\n", "v_mul_f32 %11, %11, %14\nv_mul_f32 %8, %8, %14\nv_mul_f32 %12, %12, %14\n", "v_mul_f32 %9, %9, %14\nv_mul_f32 %13, %13, %14\nv_mul_f32 %6, %6, %14\n", "v_mul_f32 %10, %10, %14\nv_mul_f32 %7, %7, %14\nv_mul_f32 %11, %11, %14\n")
KERNEL(k_mul4, "v_mul_f32 %6, %6, %14\nv_mul_f32 %10, %10, %14\nv_mul_f32 %7, %7, %14\nv_mul_f32 %11, %11, %14\n", "v_mul_f32 %8, %8, %14\nv_mul_f32 %12, %12, %14\nv_mul_f32 %9, %9, %14\nv_mul_f32 %13, %13, %14\n", "v_mul_f32 %6, %6, %14\nv_mul_f32 %10, %10, %14\nv_mul_f32 %7, %7, %14\nv_mul_f32 %11, %11, %14\n", "v_mul_f32 %8, %8, %14\nv_mul_f32 %12, %12, %14\nv_mul_f32 %9, %9, %14\nv_mul_f32 %13, %13, %14\n")
KERNEL(k_mul6, "v_mul_f32 %6, %6, %14\nv_mul_f32 %10, %10, %14\nv_mul_f32 %7, %7, %14\nv_mul_f32 %11, %11, %14\nv_mul_f32 %8, %8, %14\nv_mul_f32 %12, %12, %14\n", "v_mul_f32 %9, %9, %14\nv_mul_f32 %13, %13, %14\nv_mul_f32 %6, %6, %14\nv_mul_f32 %10, %10, %14\nv_mul_f32 %7, %7, %14\nv_mul_f32 %11, %11, %14\n", "v_mul_f32 %8, %8, %14\nv_mul_f32 %12, %12, %14\nv_mul_f32 %9, %9, %14\nv_mul_f32 %13, %13, %14\nv_mul_f32 %6, %6, %14\nv_mul_f32 %10, %10, %14\n", "v_mul_f32 %7, %7, %14\nv_mul_f32 %11, %11, %14\nv_mul_f32 %8, %8, %14\nv_mul_f32 %12, %12, %14\nv_mul_f32 %9, %9, %14\nv_mul_f32 %13, %13, %14\n")
KERNEL(k_mixlo1, "v_fma_mixlo_f16 %6, %6, %14, 0\n", "v_fma_mixlo_f16 %7, %7, %14, 0\n", "v_fma_mixlo_f16 %8, %8, %14, 0\n", "v_fma_mixlo_f16 %9, %9, %14, 0\n")
KERNEL(k_mixlo2, "v_fma_mixlo_f16 %6, %6, %14, 0\nv_fma_mixlo_f16 %10, %10, %14, 0\n", "v_fma_mixlo_f16 %7, %7, %14, 0\nv_fma_mixlo_f16 %11, %11, %14, 0\n", "v_fma_mixlo_f16 %8, %8, %14, 0\nv_fma_mixlo_f16 %12, %12, %14, 0\n", "v_fma_mixlo_f16 %9, %9, %14, 0\nv_fma_mixlo_f16 %13, %13, %14, 0\n")
KERNEL(k_cvt1, "v_cvt_pk_f16_f32 %6, %6, %14\n", "v_cvt_pk_f16_f32 %7, %7, %14\n", "v_cvt_pk_f16_f32 %8, %8, %14\n", "v_cvt_pk_f16_f32 %9, %9, %14\n")
KERNEL(k_cvt2, "v_cvt_pk_f16_f32 %6, %6, %14\nv_cvt_pk_f16_f32 %10, %10, %14\n", "v_cvt_pk_f16_f32 %7, %7, %14\nv_cvt_pk_f16_f32 %11, %11, %14\n", "v_cvt_pk_f16_f32 %8, %8, %14\nv_cvt_pk_f16_f32 %12, %12, %14\n", "v_cvt_pk_f16_f32 %9, %9, %14\nv_cvt_pk_f16_f32 %13, %13, %14\n")
KERNEL(k_mixf1, "v_fma_mix_f32 %6, %6, %14, %6\n", "v_fma_mix_f32 %7, %7, %14, %7\n", "v_fma_mix_f32 %8, %8, %14, %8\n", "v_fma_mix_f32 %9, %9, %14, %9\n")
KERNEL(k_pkmul1, "v_pk_mul_f32 %16, %16, %15\n", "v_pk_mul_f32 %17, %17, %15\n", "v_pk_mul_f32 %18, %18, %15\n", "v_pk_mul_f32 %19, %19, %15\n")
KERNEL(k_pkmul2, "v_pk_mul_f32 %16, %16, %15\nv_pk_mul_f32 %18, %18, %15\n", "v_pk_mul_f32 %17, %17, %15\nv_pk_mul_f32 %19, %19, %15\n", "v_pk_mul_f32 %16, %16, %15\nv_pk_mul_f32 %18, %18, %15\n", "v_pk_mul_f32 %17, %17, %15\nv_pk_mul_f32 %19, %19, %15\n")
KERNEL(k_sqrt1, "v_sqrt_f32 %6, %6\n", "v_sqrt_f32 %7, %7\n", "v_sqrt_f32 %8, %8\n", "v_sqrt_f32 %9, %9\n")
KERNEL(k_dswr1, "ds_write_b64 %20, %16\n", "ds_write_b64 %20, %17 offset:8\n", "ds_write_b64 %20, %18\n", "ds_write_b64 %20, %19 offset:8\n")
KERNEL(k_dsrd1, "ds_read_b64 %16, %20\n", "ds_read_b64 %17, %20 offset:8\n", "ds_read_b64 %18, %20\n", "ds_read_b64 %19, %20 offset:8\n")
KERNEL(k_salu2, "s_add_u32 s20, s20, 1\ns_add_u32 s21, s21, 1\n", "s_add_u32 s20, s20, 1\ns_add_u32 s21, s21, 1\n", "s_add_u32 s20, s20, 1\ns_add_u32 s21, s21, 1\n", "s_add_u32 s20, s20, 1\ns_add_u32 s21, s21, 1\n")
KERNEL(k_nop0_1, "s_nop 0\n", "s_nop 0\n", "s_nop 0\n", "s_nop 0\n")
KERNEL(k_mul1_salu1, "v_mul_f32 %6, %6, %14\ns_add_u32 s20, s20, 1\n", "v_mul_f32 %7, %7, %14\ns_add_u32 s20, s20, 1\n", "v_mul_f32 %8, %8, %14\ns_add_u32 s20, s20, 1\n", "v_mul_f32 %9, %9, %14\ns_add_u32 s20, s20, 1\n")


KERNEL(k_wr_then_mul, "ds_write_b64 %20, %16\nv_mul_f32 %6, %6, %14\n", "ds_write_b64 %20, %17 offset:8\nv_mul_f32 %7, %7, %14\n", "ds_write_b64 %20, %18\nv_mul_f32 %8, %8, %14\n", "ds_write_b64 %20, %19 offset:8\nv_mul_f32 %9, %9, %14\n")
KERNEL(k_mul_then_wr, "v_mul_f32 %6, %6, %14\nds_write_b64 %20, %16\n", "v_mul_f32 %7, %7, %14\nds_write_b64 %20, %17 offset:8\n", "v_mul_f32 %8, %8, %14\nds_write_b64 %20, %18\n", "v_mul_f32 %9, %9, %14\nds_write_b64 %20, %19 offset:8\n")
KERNEL(k_mul2_then_wr, "v_mul_f32 %6, %6, %14\nv_mul_f32 %10, %10, %14\nds_write_b64 %20, %16\n", "v_mul_f32 %7, %7, %14\nv_mul_f32 %11, %11, %14\nds_write_b64 %20, %17 offset:8\n", "v_mul_f32 %8, %8, %14\nv_mul_f32 %12, %12, %14\nds_write_b64 %20, %18\n", "v_mul_f32 %9, %9, %14\nv_mul_f32 %13, %13, %14\nds_write_b64 %20, %19 offset:8\n")
KERNEL(k_dswr32, "ds_write_b32 %20, %6\n", "ds_write_b32 %20, %7 offset:8\n", "ds_write_b32 %20, %8\n", "ds_write_b32 %20, %9 offset:8\n")
KERNEL(k_dswr_half, "ds_write_b64 %20, %16\n", "", "ds_write_b64 %20, %18\n", "")
KERNEL(k_mul_then_pk, "v_mul_f32 %6, %6, %14\nv_pk_mul_f32 %16, %16, %15\n", "v_mul_f32 %7, %7, %14\nv_pk_mul_f32 %17, %17, %15\n", "v_mul_f32 %8, %8, %14\nv_pk_mul_f32 %18, %18, %15\n", "v_mul_f32 %9, %9, %14\nv_pk_mul_f32 %19, %19, %15\n")
KERNEL(k_mul2_then_pk, "v_mul_f32 %6, %6, %14\nv_mul_f32 %10, %10, %14\nv_pk_mul_f32 %16, %16, %15\n", "v_mul_f32 %7, %7, %14\nv_mul_f32 %11, %11, %14\nv_pk_mul_f32 %17, %17, %15\n", "v_mul_f32 %8, %8, %14\nv_mul_f32 %12, %12, %14\nv_pk_mul_f32 %18, %18, %15\n", "v_mul_f32 %9, %9, %14\nv_mul_f32 %13, %13, %14\nv_pk_mul_f32 %19, %19, %15\n")
KERNEL(k_pkadd1, "v_pk_add_f32 %16, %16, %15\n", "v_pk_add_f32 %17, %17, %15\n", "v_pk_add_f32 %18, %18, %15\n", "v_pk_add_f32 %19, %19, %15\n")
KERNEL(k_swap1, "v_permlane32_swap_b32 %6, %10\n", "v_permlane32_swap_b32 %7, %11\n", "v_permlane32_swap_b32 %8, %12\n", "v_permlane32_swap_b32 %9, %13\n")
KERNEL(k_max3_1, "v_max3_f32 %6, %6, %14, %10\n", "v_max3_f32 %7, %7, %14, %11\n", "v_max3_f32 %8, %8, %14, %12\n", "v_max3_f32 %9, %9, %14, %13\n")
KERNEL(k_accrd1, "v_accvgpr_read_b32 %6, a255\n", "v_accvgpr_read_b32 %7, a255\n", "v_accvgpr_read_b32 %8, a255\n", "v_accvgpr_read_b32 %9, a255\n")
KERNEL(k_rfl1, "v_readfirstlane_b32 s20, %6\n", "v_readfirstlane_b32 s20, %7\n", "v_readfirstlane_b32 s20, %8\n", "v_readfirstlane_b32 s20, %9\n")
KERNEL(k_dsrd128, "ds_read_b128 %21, %20\n", "", "ds_read_b128 %21, %20 offset:16\n", "")
KERNEL(k_dsrd128_wait, "ds_read_b128 %21, %20\n", "", "s_waitcnt lgkmcnt(0)\n", "")
KERNEL(k_dswr128, "ds_write_b128 %20, %21\n", "", "ds_write_b128 %20, %21 offset:16\n", "")
KERNEL(k_vmem1, "buffer_load_dwordx4 %21, %22, s[24:27], 0 offen\n", "", "", "")
KERNEL(k_mixlo_mul, "v_fma_mixlo_f16 %6, %6, %14, 0\nv_mul_f32 %10, %10, %14\n", "v_fma_mixlo_f16 %7, %7, %14, 0\nv_mul_f32 %11, %11, %14\n", "v_fma_mixlo_f16 %8, %8, %14, 0\nv_mul_f32 %12, %12, %14\n", "v_fma_mixlo_f16 %9, %9, %14, 0\nv_mul_f32 %13, %13, %14\n")
KERNEL(k_mul_mixlo, "v_mul_f32 %10, %10, %14\nv_fma_mixlo_f16 %6, %6, %14, 0\n", "v_mul_f32 %11, %11, %14\nv_fma_mixlo_f16 %7, %7, %14, 0\n", "v_mul_f32 %12, %12, %14\nv_fma_mixlo_f16 %8, %8, %14, 0\n", "v_mul_f32 %13, %13, %14\nv_fma_mixlo_f16 %9, %9, %14, 0\n")


// the same with eight accumulators round robin (dependent MFMAs eight apart) and with A architectural
#define MFMA8(m) "v_mfma_f32_16x16x32_f16 %" #m ", %8, %9, %" #m "\n"
#define KERNEL8(NAME, ACLS, X)                                                                                             \
    __global__ void __launch_bounds__(256) NAME(int iters, unsigned long long *cyc, float *sink)                           \
    {                                                                                                                      \
        const int lane = threadIdx.x & 63;                                                                                 \
        floatx4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;                           \
        uint32x4 A = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, B = A;                                            \
        float r0 = lane, r1 = lane + 1, r2 = lane + 2, r3 = lane + 3, c = 1.0000001f;                                       \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                        \
        for (int it = 0; it < iters; it++) {                                                                               \
            asm volatile(MFMA8(0) X MFMA8(1) X MFMA8(2) X MFMA8(3) X MFMA8(4) X MFMA8(5) X MFMA8(6) X MFMA8(7) X            \
                         : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+a"(c4), "+a"(c5), "+a"(c6), "+a"(c7)                  \
                         : ACLS(A), "v"(B), "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(c));                                   \
        }                                                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                        \
        if (lane == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;                                                   \
        float s = c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0];                                           \
        if (s == 12345.678f) sink[0] = s + r0;                                                                             \
    }
KERNEL8(k8_none, "a", "")
KERNEL8(k8_none_va, "v", "")
KERNEL8(k8_mul2, "a", "v_mul_f32 %10, %10, %14\nv_mul_f32 %11, %11, %14\n")
KERNEL8(k8_mul4, "a", "v_mul_f32 %10, %10, %14\nv_mul_f32 %11, %11, %14\nv_mul_f32 %12, %12, %14\nv_mul_f32 %13, %13, %14\n")

struct Case { const char *name; void (*fn)(int, unsigned long long *, float *); };

int main()
{
    const Case cases[] = {
        {"MFMA alone", k_none}, {"MFMA alone, 8 accumulators", k8_none}, {"MFMA alone, 8 acc, A architectural", k8_none_va}, {"8 acc + 2 v_mul_f32", k8_mul2}, {"8 acc + 4 v_mul_f32", k8_mul4}, {"+ 1 v_mul_f32", k_mul1}, {"+ 2 v_mul_f32", k_mul2}, {"+ 3 v_mul_f32", k_mul3}, {"+ 4 v_mul_f32", k_mul4}, {"+ 6 v_mul_f32", k_mul6},
        {"+ 1 v_fma_mixlo_f16", k_mixlo1}, {"+ 2 v_fma_mixlo_f16", k_mixlo2}, {"+ 1 v_cvt_pk_f16_f32", k_cvt1}, {"+ 2 v_cvt_pk_f16_f32", k_cvt2},
        {"+ 1 v_fma_mix_f32", k_mixf1}, {"+ 1 v_pk_mul_f32", k_pkmul1}, {"+ 2 v_pk_mul_f32", k_pkmul2}, {"+ 1 v_sqrt_f32", k_sqrt1},
        {"+ 1 ds_write_b64", k_dswr1}, {"+ 1 ds_read_b64", k_dsrd1}, {"+ 2 s_add_u32", k_salu2}, {"+ 1 s_nop 0", k_nop0_1}, {"+ 1 v_mul_f32 + 1 s_add_u32", k_mul1_salu1},
        {"+ ds_write_b64, v_mul", k_wr_then_mul}, {"+ v_mul, ds_write_b64", k_mul_then_wr}, {"+ 2 v_mul, ds_write_b64", k_mul2_then_wr},
        {"+ 1 ds_write_b32", k_dswr32}, {"+ ds_write_b64 every other tick", k_dswr_half}, {"+ v_mul, v_pk_mul_f32", k_mul_then_pk}, {"+ 2 v_mul, v_pk_mul_f32", k_mul2_then_pk},
        {"+ 1 v_pk_add_f32", k_pkadd1}, {"+ 1 v_permlane32_swap", k_swap1}, {"+ 1 v_max3_f32", k_max3_1}, {"+ 1 v_accvgpr_read", k_accrd1}, {"+ 1 v_readfirstlane", k_rfl1},
        {"+ ds_read_b128 every other tick", k_dsrd128}, {"+ ds_read_b128, waitcnt 2 ticks on", k_dsrd128_wait}, {"+ ds_write_b128 every other tick", k_dswr128},
        {"+ buffer_load_dwordx4 every 4th tick", k_vmem1}, {"+ v_fma_mixlo, v_mul", k_mixlo_mul}, {"+ v_mul, v_fma_mixlo", k_mul_mixlo},
    };
    unsigned long long *cyc; float *sink;
    CK(hipMalloc(&cyc, 256 * 4 * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, 4));
    const int iters = 4000;
    printf("%-36s %s\n", "behind every MFMA (lone wave)", "shader clocks per tick");
    for (const Case &c : cases) {
        hipLaunchKernelGGL(c.fn, dim3(256), dim3(256), 0, 0, 100, cyc, sink);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(c.fn, dim3(256), dim3(256), 0, 0, iters, cyc, sink);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(1024);
        CK(hipMemcpy(h.data(), cyc, 1024 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= 1024;
        printf("%-36s %8.2f\n", c.name, mean / ((double)iters * 8));
        fflush(stdout);
    }
    return 0;
}
