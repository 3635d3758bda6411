// Issue cost of single vector instructions on gfx950 (diagnostic, not part of the product; cited in DESIGN.md §6.0).
// Each kernel runs one instruction kind over eight independent destinations, 64 instructions an iteration, with 1, 2 or 3
// waves per SIMD; prints shader-clock cycles per instruction per SIMD (s_memtime) and the clock the run held.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rates tools/ubench/valu_rates.hip && gpurun -- ./tools/ubench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define OPS32 "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
#define OPS64 "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)

#define KERNEL(NAME, BODY, OPS)                                                                                           \
    __global__ void __launch_bounds__(1024) NAME(int iters, unsigned long long *cyc, float *sink)                         \
    {                                                                                                                     \
        const int lane = threadIdx.x & 63;                                                                                \
        float r0 = lane, r1 = lane + 1, r2 = lane + 2, r3 = lane + 3, r4 = lane + 4, r5 = lane + 5, r6 = lane + 6, r7 = lane + 7; \
        f2 p0 = {r0, r1}, p1 = {r1, r2}, p2 = {r2, r3}, p3 = {r3, r4}, p4 = {r4, r5}, p5 = {r5, r6}, p6 = {r6, r7}, p7 = {r7, r0}; \
        float c = 1.0000001f; f2 cc = {c, c};                                                                             \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                       \
        for (int it = 0; it < iters; it++) {                                                                              \
            asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY : OPS : "v"(c), "v"(cc) : "vcc", "s10", "s11", "s12", "s13", "a0", "a1", "a2", "a3");                                \
        }                                                                                                                 \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                       \
        if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;                                  \
        float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + p0.y + p7.y; \
        if (s == 12345.678f) sink[0] = s;                                                                                 \
    }

// operands: %0..%7 the destinations, %8 a scalar-ish constant, %9 a packed constant
#define B32(INS) INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n"
#define B32_3(INS) INS " %0, %0, %8, %0\n" INS " %1, %1, %8, %1\n" INS " %2, %2, %8, %2\n" INS " %3, %3, %8, %3\n" INS " %4, %4, %8, %4\n" INS " %5, %5, %8, %5\n" INS " %6, %6, %8, %6\n" INS " %7, %7, %8, %7\n"
#define B32_1(INS, SUF) INS " %0, %0" SUF "\n" INS " %1, %1" SUF "\n" INS " %2, %2" SUF "\n" INS " %3, %3" SUF "\n" INS " %4, %4" SUF "\n" INS " %5, %5" SUF "\n" INS " %6, %6" SUF "\n" INS " %7, %7" SUF "\n"
#define B32_X(INS) INS " %0, %1\n" INS " %2, %3\n" INS " %4, %5\n" INS " %6, %7\n" INS " %1, %2\n" INS " %3, %4\n" INS " %5, %6\n" INS " %7, %0\n"
#define B64(INS) INS " %0, %0, %9\n" INS " %1, %1, %9\n" INS " %2, %2, %9\n" INS " %3, %3, %9\n" INS " %4, %4, %9\n" INS " %5, %5, %9\n" INS " %6, %6, %9\n" INS " %7, %7, %9\n"
#define B64_3(INS) INS " %0, %0, %9, %0\n" INS " %1, %1, %9, %1\n" INS " %2, %2, %9, %2\n" INS " %3, %3, %9, %3\n" INS " %4, %4, %9, %4\n" INS " %5, %5, %9, %5\n" INS " %6, %6, %9, %6\n" INS " %7, %7, %9, %7\n"
#define B64_SW(INS) INS " %0, %0, %9, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n" INS " %1, %1, %9, %1 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n" INS " %2, %2, %9, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n" INS " %3, %3, %9, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n" INS " %4, %4, %9, %4 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n" INS " %5, %5, %9, %5 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n" INS " %6, %6, %9, %6 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n" INS " %7, %7, %9, %7 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n"

KERNEL(k_mul, B32("v_mul_f32"), OPS32)
KERNEL(k_add, B32("v_add_f32"), OPS32)
KERNEL(k_fma, B32_3("v_fma_f32"), OPS32)
KERNEL(k_and, B32("v_and_b32"), OPS32)
KERNEL(k_mov, B32_1("v_mov_b32", ""), OPS32)
KERNEL(k_pk_mul, B64("v_pk_mul_f32"), OPS64)
KERNEL(k_pk_add, B64("v_pk_add_f32"), OPS64)
KERNEL(k_pk_fma, B64_3("v_pk_fma_f32"), OPS64)
KERNEL(k_pk_fma_swz, B64_SW("v_pk_fma_f32"), OPS64)
KERNEL(k_pk_mov, B64("v_pk_mov_b32"), OPS64)
KERNEL(k_fma_mix, B32_3("v_fma_mix_f32"), OPS32)
KERNEL(k_fma_mixlo, B32_3("v_fma_mixlo_f16"), OPS32)
KERNEL(k_cvt_pkrtz, B32("v_cvt_pkrtz_f16_f32"), OPS32)
KERNEL(k_cvt_pk_f16, B32("v_cvt_pk_f16_f32"), OPS32)
KERNEL(k_cvt_f16, B32_1("v_cvt_f16_f32", ""), OPS32)
KERNEL(k_pk_add_f16, B32("v_pk_add_f16"), OPS32)
KERNEL(k_pk_fma_f16, B32_3("v_pk_fma_f16"), OPS32)
KERNEL(k_sqrt, B32_1("v_sqrt_f32", ""), OPS32)
KERNEL(k_rsq, B32_1("v_rsq_f32", ""), OPS32)
KERNEL(k_exp, B32_1("v_exp_f32", ""), OPS32)
KERNEL(k_rcp, B32_1("v_rcp_f32", ""), OPS32)
KERNEL(k_perm, B32_3("v_perm_b32"), OPS32)
#define B32_S(INS, SUF) INS " %0, %0, %8" SUF "\n" INS " %1, %1, %8" SUF "\n" INS " %2, %2, %8" SUF "\n" INS " %3, %3, %8" SUF "\n" INS " %4, %4, %8" SUF "\n" INS " %5, %5, %8" SUF "\n" INS " %6, %6, %8" SUF "\n" INS " %7, %7, %8" SUF "\n"
KERNEL(k_cndmask, B32_S("v_cndmask_b32", ", vcc"), OPS32)
KERNEL(k_cndmask_sgpr, B32_S("v_cndmask_b32_e64", ", s[10:11]"), OPS32)
#define B32_CI(INS) INS " %0, 0, %0, vcc\n" INS " %1, 0, %1, vcc\n" INS " %2, 0, %2, vcc\n" INS " %3, 0, %3, vcc\n" INS " %4, 0, %4, vcc\n" INS " %5, 0, %5, vcc\n" INS " %6, 0, %6, vcc\n" INS " %7, 0, %7, vcc\n"
KERNEL(k_cndmask_imm, B32_CI("v_cndmask_b32_e64"), OPS32)
#define B32_D(INS) INS " %0, %1, %8, vcc\n" INS " %1, %2, %8, vcc\n" INS " %2, %3, %8, vcc\n" INS " %3, %4, %8, vcc\n" INS " %4, %5, %8, vcc\n" INS " %5, %6, %8, vcc\n" INS " %6, %7, %8, vcc\n" INS " %7, %0, %8, vcc\n"
KERNEL(k_cndmask_chain, B32_D("v_cndmask_b32"), OPS32)
#define B_SAND_SEL "s_and_b64 vcc, s[10:11], s[12:13]\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n s_and_b64 vcc, s[10:11], s[12:13]\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n"
KERNEL(k_sand_sel, B_SAND_SEL, OPS32)
#define B_CMP1_SEL7 "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
KERNEL(k_cmp1_sel7, B_CMP1_SEL7, OPS32)
#define B_CMP1_FMA_SEL "v_cmp_lt_f32 vcc, %0, %8\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_cndmask_b32 %4, %4, %8, vcc\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_cndmask_b32 %7, %7, %8, vcc\n"
KERNEL(k_cmp1_fma_sel, B_CMP1_FMA_SEL, OPS32)
KERNEL(k_bfi, B32_3("v_bfi_b32"), OPS32)
KERNEL(k_max3, B32_3("v_max3_f32"), OPS32)
KERNEL(k_mul_lo, B32("v_mul_lo_u32"), OPS32)
KERNEL(k_mad_u24, B32_3("v_mad_u32_u24"), OPS32)
KERNEL(k_add_u32, B32("v_add_u32"), OPS32)
#define B32_CMP(INS) INS " vcc, %0, %8\n" INS " vcc, %1, %8\n" INS " vcc, %2, %8\n" INS " vcc, %3, %8\n" INS " vcc, %4, %8\n" INS " vcc, %5, %8\n" INS " vcc, %6, %8\n" INS " vcc, %7, %8\n"
KERNEL(k_cmp_f32, B32_CMP("v_cmp_lt_f32"), OPS32)
#define B64_CMP(INS) INS " vcc, %0, %9\n" INS " vcc, %1, %9\n" INS " vcc, %2, %9\n" INS " vcc, %3, %9\n" INS " vcc, %4, %9\n" INS " vcc, %5, %9\n" INS " vcc, %6, %9\n" INS " vcc, %7, %9\n"
KERNEL(k_cmp_f64, B64_CMP("v_cmp_le_f64"), OPS64)
#define B32_CMPSEL "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
KERNEL(k_cmp_sel, B32_CMPSEL, OPS32)
#define B_ACCRW "v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_write_b32 a3, %3\n v_accvgpr_read_b32 %4, a0\n v_accvgpr_read_b32 %5, a1\n v_accvgpr_read_b32 %6, a2\n v_accvgpr_read_b32 %7, a3\n"
KERNEL(k_accrw, B_ACCRW, OPS32)
#define B_NOP0 "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
#define B_NOP3 "s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n"
#define B_SMOV "s_mov_b32 s10, s11\n s_mov_b32 s12, s13\n s_mov_b32 s10, s11\n s_mov_b32 s12, s13\n s_mov_b32 s10, s11\n s_mov_b32 s12, s13\n s_mov_b32 s10, s11\n s_mov_b32 s12, s13\n"
KERNEL(k_nop0, B_NOP0, OPS32)
KERNEL(k_nop3, B_NOP3, OPS32)
KERNEL(k_smov, B_SMOV, OPS32)
#define B_RFL "v_readfirstlane_b32 s10, %0\n v_readfirstlane_b32 s11, %1\n v_readfirstlane_b32 s12, %2\n v_readfirstlane_b32 s13, %3\n v_readfirstlane_b32 s10, %4\n v_readfirstlane_b32 s11, %5\n v_readfirstlane_b32 s12, %6\n v_readfirstlane_b32 s13, %7\n"
KERNEL(k_rfl, B_RFL, OPS32)
#define B_CVT64 "v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %8\n v_cvt_f64_f32 %2, %8\n v_cvt_f64_f32 %3, %8\n v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %8\n v_cvt_f64_f32 %6, %8\n v_cvt_f64_f32 %7, %8\n"
KERNEL(k_cvt_f64, B_CVT64, OPS64)
KERNEL(k_lshl_add, B32_3("v_lshl_add_u32"), OPS32)
KERNEL(k_mov_dpp_quad, B32_1("v_mov_b32_dpp", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"), OPS32)
KERNEL(k_mov_dpp_rowshr, B32_1("v_mov_b32_dpp", " row_shr:4 row_mask:0xf bank_mask:0xa"), OPS32)
KERNEL(k_mov_dpp_rowror, B32_1("v_mov_b32_dpp", " row_ror:8 row_mask:0xf bank_mask:0xf"), OPS32)
KERNEL(k_add_dpp, B32_S("v_add_f32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf"), OPS32)
KERNEL(k_permlane32_swap, B32_X("v_permlane32_swap_b32"), OPS32)
KERNEL(k_permlane16_swap, B32_X("v_permlane16_swap_b32"), OPS32)
KERNEL(k_swap, B32_X("v_swap_b32"), OPS32)

struct Case { const char *name; void (*fn)(int, unsigned long long *, float *); };

int main()
{
    const Case cases[] = {
        {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_fma_f32", k_fma}, {"v_and_b32", k_and}, {"v_mov_b32", k_mov},
        {"v_pk_mul_f32", k_pk_mul}, {"v_pk_add_f32", k_pk_add}, {"v_pk_fma_f32", k_pk_fma}, {"v_pk_fma_f32 op_sel/neg", k_pk_fma_swz},
        {"v_pk_mov_b32", k_pk_mov},
        {"v_fma_mix_f32", k_fma_mix}, {"v_fma_mixlo_f16", k_fma_mixlo}, {"v_cvt_pkrtz_f16_f32", k_cvt_pkrtz}, {"v_cvt_pk_f16_f32", k_cvt_pk_f16},
        {"v_cvt_f16_f32", k_cvt_f16}, {"v_pk_add_f16", k_pk_add_f16}, {"v_pk_fma_f16", k_pk_fma_f16},
        {"v_sqrt_f32", k_sqrt}, {"v_rsq_f32", k_rsq}, {"v_exp_f32", k_exp}, {"v_rcp_f32", k_rcp},
        {"v_perm_b32", k_perm}, {"v_cndmask_b32", k_cndmask}, {"v_lshl_add_u32", k_lshl_add},
        {"v_cndmask_b32_e64 s[10:11]", k_cndmask_sgpr}, {"v_cndmask_b32_e64 0, v, vcc", k_cndmask_imm}, {"v_cndmask_b32 d!=src", k_cndmask_chain},
        {"s_and vcc + 3 cndmask_e32", k_sand_sel}, {"1 v_cmp + 7 cndmask_e32", k_cmp1_sel7}, {"v_cmp, 3 fma, sel, 2 fma, sel", k_cmp1_fma_sel}, {"v_bfi_b32", k_bfi}, {"v_max3_f32", k_max3}, {"v_mul_lo_u32", k_mul_lo}, {"v_mad_u32_u24", k_mad_u24}, {"v_add_u32", k_add_u32},
        {"v_cmp_lt_f32 vcc", k_cmp_f32}, {"v_cmp_le_f64 vcc", k_cmp_f64}, {"v_cmp + v_cndmask pairs", k_cmp_sel}, {"v_accvgpr write/read", k_accrw},
        {"s_nop 0", k_nop0}, {"s_nop 3", k_nop3}, {"s_mov_b32", k_smov}, {"v_readfirstlane_b32", k_rfl}, {"v_cvt_f64_f32", k_cvt_f64},
        {"v_mov_b32_dpp quad_perm", k_mov_dpp_quad}, {"v_mov_b32_dpp row_shr bank_mask", k_mov_dpp_rowshr}, {"v_mov_b32_dpp row_ror", k_mov_dpp_rowror},
        {"v_add_f32_dpp row_shr:1", k_add_dpp},
        {"v_permlane32_swap_b32", k_permlane32_swap}, {"v_permlane16_swap_b32", k_permlane16_swap}, {"v_swap_b32", k_swap},
    };
    unsigned long long *cyc; float *sink;
    CK(hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, 4));
    const int iters = 4000;
    printf("%-40s %28s %28s %28s\n", "instruction", "1 wave/SIMD cyc/instr (GHz)", "2 waves/SIMD", "3 waves/SIMD");
    for (const Case &c : cases) {
        printf("%-40s", c.name);
        for (int w = 1; w <= 3; w++) {
            const int block = 256 * w, nw = 256 * 4 * w;
            hipLaunchKernelGGL(c.fn, dim3(256), dim3(block), 0, 0, 200, cyc, sink);
            CK(hipDeviceSynchronize());
            const auto h0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(c.fn, dim3(256), dim3(block), 0, 0, iters, cyc, sink);
            CK(hipDeviceSynchronize());
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count();
            std::vector<unsigned long long> h(nw);
            CK(hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double mean = 0;
            for (auto v : h) mean += (double)v;
            mean /= nw;
            // s_memtime counts at a fixed 100 MHz on this part; wall clock over the launch gives the time, per SIMD instruction slots follow
            const double instr_per_simd = (double)iters * 64 * w;
            printf("   %8.2f ns/instr (memtime %6.2f)", wall * 1e9 / instr_per_simd, mean / instr_per_simd);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
