// pkfma_opsel.hip -- does a packed fp32 instruction keep both halves of its result in every lane under every operand selection,
// whatever the SIMD's other waves are doing?  Round 5's per-chunk trace of the wide GEMM (tools/debug/wide_trace.py) found the
// staggered form's wrong blocks to be ONE multiply-add missing from evaluation tile 0's running sum (the low half of a register
// pair) in lanes 48-63, always the term the compiler wrote as  v_pk_fma_f32 ... op_sel:[0,1,0]  (both halves take the HIGH
// register of the weight pair), while the SIMD's partner wave was inside its matrix instructions.  This program runs that
// instruction in isolation: "vector" waves (4-7 of a 512-thread workgroup, two workgroups a CU) execute four packed instructions
// separated by a chosen filler (one assembly block: the adjacency is exact) and check both halves of every result against plain
// v_fma_f32 / v_mul_f32 / v_add_f32; their SIMD partners (waves 0-3) idle, run v_mfma_f32_16x16x32_bf16, or run v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pkfma_opsel tools/ubench/pkfma_opsel.hip && gpurun -- ./tools/ubench/pkfma_opsel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void partner_work(int partner, int iters, int *done, float *sink, int lane)
{
    if (partner == 1) {
        floatx4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        bf16x8 a8, b8;
        for (int i = 0; i < 8; i++) { a8[i] = (__bf16)(lane * 0.001f + i); b8[i] = (__bf16)(i * 0.5f); }
        // (a bounded loop, about as long as the vector waves' -- and it leaves early when they are done: nothing here can wait for ever)
        for (int it = 0; it < iters && __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4; it++) {
#pragma unroll
            for (int q = 0; q < 32; q++) acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[q & 3], 0, 0, 0);
        }
        sink[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else if (partner == 2) {
        float v[8];
        for (int i = 0; i < 8; i++) v[i] = lane * 0.01f + i;
        for (int it = 0; it < iters && __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4; it++) {
#pragma unroll
            for (int q = 0; q < 64; q++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(0.999f), "v"(0.001f));
        }
        sink[blockIdx.x * 512 + threadIdx.x] = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
    }
}

// counts[0..3]: wrong low halves by lane quarter; [4..7]: wrong high halves; [8]: wrong low halves that equal the addend (the product
// is gone); [9]: wrong low halves that equal the result under NO selection (the selection was ignored)
// INSN(i): the packed instruction on y (in/out), h<i>, w<i>;  LO / HI(h, w, y): what each half must be;  NOSEL(h, w, y): the low half without selection
#define TEST_KERNEL(NAME, SEP, INSN, LO, HI, NOSEL)                                                                                        \
    __global__ void __launch_bounds__(512, 4) NAME(unsigned *counts, float *sink, int iters, int partner)                                  \
    {                                                                                                                                       \
        __shared__ int done;                                                                                                                \
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                                                         \
        if (threadIdx.x == 0) done = 0;                                                                                                     \
        __syncthreads();                                                                                                                    \
        if (wave < 4) { partner_work(partner, iters, &done, sink, lane); return; }                                                          \
        floatx2 y = {0.25f + lane * 0.001f, -0.5f + lane * 0.002f};                                                                         \
        floatx2 h[4], w[4];                                                                                                                 \
        float x[8], r[8];                                                                                                                   \
        for (int i = 0; i < 4; i++) {                                                                                                       \
            h[i] = floatx2{0.3f + 0.01f * i + lane * 0.0003f, 0.6f - 0.02f * i + lane * 0.0005f};                                           \
            w[i] = floatx2{(i & 1) ? 0.031f : -0.029f, (i & 1) ? -0.027f : 0.033f};                                                        \
        }                                                                                                                                   \
        for (int i = 0; i < 8; i++) x[i] = 1.5f + 0.1f * i + lane * 0.01f;                                                                  \
        unsigned bad_lo = 0, bad_hi = 0, gone = 0, ignored = 0;                                                                             \
        for (int it = 0; it < iters; it++) {                                                                                                \
            floatx2 yr = y, yn = y;                                                                                                         \
            float last_in = 0.0f;                                                                                                           \
            _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                                                 \
                last_in = yr.x;                                                                                                             \
                yn.x = NOSEL(h[i], w[i], yr);                                                                                               \
                const float lo_ = LO(h[i], w[i], yr), hi_ = HI(h[i], w[i], yr);                                                             \
                yr = floatx2{lo_, hi_};                                                                                                     \
            }                                                                                                                               \
            asm volatile(SEP(0, 1) INSN(0) SEP(2, 3) INSN(1) SEP(4, 5) INSN(2) SEP(6, 7) INSN(3)                                            \
                         : [y] "+v"(y), [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3]), [r4] "=&v"(r[4]), [r5] "=&v"(r[5]), [r6] "=&v"(r[6]), [r7] "=&v"(r[7]) \
                         : [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), \
                           [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7])); \
            if (__float_as_uint(y.x) != __float_as_uint(yr.x)) {                                                                            \
                bad_lo++;                                                                                                                   \
                if (__float_as_uint(y.x) == __float_as_uint(last_in)) gone++;       /* (only the LAST of the four leaves this trace) */    \
                if (__float_as_uint(y.x) == __float_as_uint(yn.x)) ignored++;                                                               \
            }                                                                                                                               \
            if (__float_as_uint(y.y) != __float_as_uint(yr.y)) bad_hi++;                                                                    \
            y = floatx2{yr.x * 0.5f + 0.125f, yr.y * 0.5f - 0.125f};                                                                        \
            _Pragma("unroll") for (int i = 0; i < 4; i++) h[i] = floatx2{0.25f + 0.5f * (r[2 * i] - (int)r[2 * i]), 0.75f - 0.5f * (r[2 * i + 1] - (int)r[2 * i + 1])}; \
            _Pragma("unroll") for (int i = 0; i < 8; i++) x[i] = x[i] + 0.001f > 3.0f ? 1.5f : x[i] + 0.001f;                               \
        }                                                                                                                                   \
        if (bad_lo) atomicAdd(&counts[lane >> 4], bad_lo);                                                                                  \
        if (bad_hi) atomicAdd(&counts[4 + (lane >> 4)], bad_hi);                                                                            \
        if (gone) atomicAdd(&counts[8], gone);                                                                                              \
        if (ignored) atomicAdd(&counts[9], ignored);                                                                                        \
        if (lane == 0) __hip_atomic_fetch_add(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                                    \
        sink[blockIdx.x * 512 + threadIdx.x] = y.x + y.y;                                                                                   \
    }

__device__ __forceinline__ float fma_(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float mul_(float a, float b) { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float add_(float a, float b) { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

// fillers between the packed instructions (they write r<a>, r<b> so that nothing is dead)
#define SEP_RCP(a, b) "v_rcp_f32 %[r" #a "], %[x" #a "]\n\tv_rcp_f32 %[r" #b "], %[x" #b "]\n\t"
#define SEP_MUL(a, b) "v_mul_f32 %[r" #a "], %[x" #a "], %[x" #a "]\n\tv_mul_f32 %[r" #b "], %[x" #b "], %[x" #b "]\n\t"
#define SEP_NOP(a, b) "v_mov_b32 %[r" #a "], %[x" #a "]\n\tv_mov_b32 %[r" #b "], %[x" #b "]\n\ts_nop 1\n\t"
#define SEP_NOP7(a, b) "v_mov_b32 %[r" #a "], %[x" #a "]\n\tv_mov_b32 %[r" #b "], %[x" #b "]\n\ts_nop 7\n\t"
// the instruction under test
#define FMA_S1HI(i) "v_pk_fma_f32 %[y], %[h" #i "], %[w" #i "], %[y] op_sel:[0,1,0]\n\t"
#define FMA_S1LO(i) "v_pk_fma_f32 %[y], %[h" #i "], %[w" #i "], %[y] op_sel_hi:[1,0,1]\n\t"
#define FMA_NONE(i) "v_pk_fma_f32 %[y], %[h" #i "], %[w" #i "], %[y]\n\t"
#define FMA_S0HI(i) "v_pk_fma_f32 %[y], %[h" #i "], %[w" #i "], %[y] op_sel:[1,0,0]\n\t"
#define FMA_S2HI(i) "v_pk_fma_f32 %[y], %[h" #i "], %[w" #i "], %[y] op_sel:[0,0,1]\n\t"
#define FMA_W0HI(i) "v_pk_fma_f32 %[y], %[w" #i "], %[h" #i "], %[y] op_sel:[1,0,0]\n\t"          /* the weight pair as src0, its high register into both halves */
#define FMA_W0HI_DY(i) "v_pk_fma_f32 %[y], %[y], %[h" #i "], %[w" #i "] op_sel:[1,0,0]\n\t"       /* D = S0 (as the kernel's  v[82:83], v[82:83], v[92:93], v[90:91]) */
#define FMA_S1HI_NOPB(i) "s_nop 0\n\tv_pk_fma_f32 %[y], %[h" #i "], %[w" #i "], %[y] op_sel:[0,1,0]\n\t"
#define FMA_S1HI_NOPA(i) "v_pk_fma_f32 %[y], %[h" #i "], %[w" #i "], %[y] op_sel:[0,1,0]\n\ts_nop 0\n\t"
#define MUL_S1HI(i) "v_pk_mul_f32 %[y], %[y], %[w" #i "] op_sel:[0,1]\n\tv_pk_add_f32 %[y], %[y], %[h" #i "]\n\t"
#define ADD_S1HI(i) "v_pk_add_f32 %[y], %[y], %[w" #i "] op_sel:[0,1]\n\tv_pk_mul_f32 %[y], %[y], %[h" #i "]\n\t"
// what the halves must be
#define LO_S1HI(H_, W_, Y_) fma_(H_.x, W_.y, Y_.x)
#define HI_S1HI(H_, W_, Y_) fma_(H_.y, W_.y, Y_.y)
#define LO_S1LO(H_, W_, Y_) fma_(H_.x, W_.x, Y_.x)
#define HI_S1LO(H_, W_, Y_) fma_(H_.y, W_.x, Y_.y)
#define LO_NONE(H_, W_, Y_) fma_(H_.x, W_.x, Y_.x)
#define HI_NONE(H_, W_, Y_) fma_(H_.y, W_.y, Y_.y)
#define LO_S0HI(H_, W_, Y_) fma_(H_.y, W_.x, Y_.x)
#define LO_S2HI(H_, W_, Y_) fma_(H_.x, W_.x, Y_.y)
#define LO_W0HI(H_, W_, Y_) fma_(W_.y, H_.x, Y_.x)
#define HI_W0HI(H_, W_, Y_) fma_(W_.y, H_.y, Y_.y)
#define LO_DY(H_, W_, Y_) fma_(Y_.y, H_.x, W_.x)
#define HI_DY(H_, W_, Y_) fma_(Y_.y, H_.y, W_.y)
#define NS_DY(H_, W_, Y_) fma_(Y_.x, H_.x, W_.x)
#define LO_MULS(H_, W_, Y_) add_(mul_(Y_.x, W_.y), H_.x)
#define HI_MULS(H_, W_, Y_) add_(mul_(Y_.y, W_.y), H_.y)
#define NS_MULS(H_, W_, Y_) add_(mul_(Y_.x, W_.x), H_.x)
#define LO_ADDS(H_, W_, Y_) mul_(add_(Y_.x, W_.y), H_.x)
#define HI_ADDS(H_, W_, Y_) mul_(add_(Y_.y, W_.y), H_.y)
#define NS_ADDS(H_, W_, Y_) mul_(add_(Y_.x, W_.x), H_.x)

TEST_KERNEL(k_s1hi_rcp, SEP_RCP, FMA_S1HI, LO_S1HI, HI_S1HI, LO_NONE)
TEST_KERNEL(k_s1hi_mul, SEP_MUL, FMA_S1HI, LO_S1HI, HI_S1HI, LO_NONE)
TEST_KERNEL(k_s1hi_nop, SEP_NOP, FMA_S1HI, LO_S1HI, HI_S1HI, LO_NONE)
TEST_KERNEL(k_s1hi_nop7, SEP_NOP7, FMA_S1HI, LO_S1HI, HI_S1HI, LO_NONE)
TEST_KERNEL(k_s1hi_nopb, SEP_MUL, FMA_S1HI_NOPB, LO_S1HI, HI_S1HI, LO_NONE)
TEST_KERNEL(k_s1hi_nopa, SEP_MUL, FMA_S1HI_NOPA, LO_S1HI, HI_S1HI, LO_NONE)
TEST_KERNEL(k_s1lo_mul, SEP_MUL, FMA_S1LO, LO_S1LO, HI_S1LO, LO_NONE)
TEST_KERNEL(k_none_mul, SEP_MUL, FMA_NONE, LO_NONE, HI_NONE, LO_NONE)
TEST_KERNEL(k_s0hi_mul, SEP_MUL, FMA_S0HI, LO_S0HI, HI_NONE, LO_NONE)
TEST_KERNEL(k_s2hi_mul, SEP_MUL, FMA_S2HI, LO_S2HI, HI_NONE, LO_NONE)
TEST_KERNEL(k_w0hi_rcp, SEP_RCP, FMA_W0HI, LO_W0HI, HI_W0HI, LO_NONE)
TEST_KERNEL(k_w0hi_mul, SEP_MUL, FMA_W0HI, LO_W0HI, HI_W0HI, LO_NONE)
TEST_KERNEL(k_w0hi_nop, SEP_NOP, FMA_W0HI, LO_W0HI, HI_W0HI, LO_NONE)
TEST_KERNEL(k_dy_rcp, SEP_RCP, FMA_W0HI_DY, LO_DY, HI_DY, NS_DY)
TEST_KERNEL(k_dy_nop, SEP_NOP, FMA_W0HI_DY, LO_DY, HI_DY, NS_DY)
TEST_KERNEL(k_s1lo_rcp, SEP_RCP, FMA_S1LO, LO_S1LO, HI_S1LO, LO_NONE)
TEST_KERNEL(k_s1lo_nop, SEP_NOP, FMA_S1LO, LO_S1LO, HI_S1LO, LO_NONE)
TEST_KERNEL(k_none_nop, SEP_NOP, FMA_NONE, LO_NONE, HI_NONE, LO_NONE)
TEST_KERNEL(k_mul_s1hi, SEP_MUL, MUL_S1HI, LO_MULS, HI_MULS, NS_MULS)
TEST_KERNEL(k_add_s1hi, SEP_MUL, ADD_S1HI, LO_ADDS, HI_ADDS, NS_ADDS)

// The wide GEMM's own sequence (the trailing waves' epilogue after the weights moved to src0; kernels_wide.hip, -DSYLDET_WIDE_X_DMATRAIL
// build, which still differed run to run): eight packed multiply-adds on one running pair, the weights' LOW registers broadcast by
// op_sel_hi:[0,1,1], their HIGH registers taken by op_sel:[1,0,0], the second one writing over its own src0, two of them back to
// back, transcendental and plain vector instructions between the others.
__global__ void __launch_bounds__(512, 4) k_kernel_sequence(unsigned *counts, float *sink, int iters, int partner)
{
    __shared__ int done;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave < 4) { partner_work(partner, iters, &done, sink, lane); return; }
    floatx2 y = {0.25f + lane * 0.001f, -0.5f + lane * 0.002f};
    floatx2 h[8];
    float x[8], r[8];
    for (int i = 0; i < 8; i++) { h[i] = floatx2{0.3f + 0.01f * i + lane * 0.0003f, 0.6f - 0.02f * i + lane * 0.0005f}; x[i] = 1.5f + 0.1f * i + lane * 0.01f; }
    unsigned bad_lo = 0, bad_hi = 0, term[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        const float wv[8] = {-0.029f + 1e-5f * (it & 15), 0.033f, 0.031f, -0.027f, 0.021f, -0.036f, -0.024f, 0.038f};   // ut0 x y z w, ut1 x y z w
        floatx2 wa01 = {wv[0], wv[1]}, wa23 = {wv[2], wv[3]}, wb01 = {wv[4], wv[5]}, wb23 = {wv[6], wv[7]};
        floatx2 m1 = {wv[3], 0.0f}, m2 = {wv[7], 0.0f};          // (the compiler's  v_mov_b32 v112, v85  /  v114, v89)
        float ylo = y.x, yhi = y.y, lo_without[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { ylo = fma_(wv[i], h[i].x, ylo); yhi = fma_(wv[i], h[i].y, yhi); }
#pragma unroll
        for (int k = 0; k < 8; k++) {                              // the low half with term k left out
            float v = y.x;
#pragma unroll
            for (int i = 0; i < 8; i++) if (i != k) v = fma_(wv[i], h[i].x, v);
            lo_without[k] = v;
        }
        floatx2 t;
        asm volatile("v_pk_fma_f32 %[t], %[wa01], %[h0], %[y] op_sel_hi:[0,1,1]\n\t"
                     "v_add_f32 %[r0], 1.0, %[x0]\n\tv_rcp_f32 %[r1], %[x1]\n\tv_rcp_f32 %[r2], %[x2]\n\t"
                     "v_pk_fma_f32 %[wa01], %[wa01], %[h1], %[t] op_sel:[1,0,0]\n\t"
                     "v_add_f32 %[r3], 1.0, %[x3]\n\tv_rcp_f32 %[r4], %[x4]\n\tv_rcp_f32 %[r5], %[x5]\n\t"
                     "v_pk_fma_f32 %[wa01], %[wa23], %[h2], %[wa01] op_sel_hi:[0,1,1]\n\t"
                     "v_rcp_f32 %[r6], %[x6]\n\t"
                     "v_pk_fma_f32 %[wa01], %[m1], %[h3], %[wa01] op_sel_hi:[0,1,1]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_fma_f32 %[wa01], %[wb01], %[h4], %[wa01] op_sel_hi:[0,1,1]\n\t"
                     "v_pk_fma_f32 %[wa01], %[wb01], %[h5], %[wa01] op_sel:[1,0,0]\n\t"
                     "v_mov_b32 %[r7], %[x7]\n\t"
                     "v_pk_fma_f32 %[wa01], %[wb23], %[h6], %[wa01] op_sel_hi:[0,1,1]\n\t"
                     "v_pk_fma_f32 %[y], %[m2], %[h7], %[wa01] op_sel_hi:[0,1,1]\n\t"
                     : [y] "+v"(y), [t] "=&v"(t), [wa01] "+v"(wa01), [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3]), [r4] "=&v"(r[4]),
                       [r5] "=&v"(r[5]), [r6] "=&v"(r[6]), [r7] "=&v"(r[7])
                     : [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]), [h3] "v"(h[3]), [h4] "v"(h[4]), [h5] "v"(h[5]), [h6] "v"(h[6]), [h7] "v"(h[7]),
                       [wa23] "v"(wa23), [wb01] "v"(wb01), [wb23] "v"(wb23), [m1] "v"(m1), [m2] "v"(m2),
                       [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]));
        if (__float_as_uint(y.x) != __float_as_uint(ylo)) {
            bad_lo++;
#pragma unroll
            for (int k = 0; k < 8; k++) if (__float_as_uint(y.x) == __float_as_uint(lo_without[k])) term[k]++;
        }
        if (__float_as_uint(y.y) != __float_as_uint(yhi)) bad_hi++;
        y = floatx2{ylo * 0.5f + 0.125f, yhi * 0.5f - 0.125f};
#pragma unroll
        for (int i = 0; i < 8; i++) { h[i] = floatx2{0.25f + 0.5f * (r[i] - (int)r[i]), 0.75f - 0.5f * (r[(i + 3) & 7] - (int)r[(i + 3) & 7])}; x[i] = x[i] + 0.001f > 3.0f ? 1.5f : x[i] + 0.001f; }
    }
    if (bad_lo) atomicAdd(&counts[lane >> 4], bad_lo);
    if (bad_hi) atomicAdd(&counts[4 + (lane >> 4)], bad_hi);
#pragma unroll
    for (int k = 0; k < 8; k++) if (term[k]) atomicAdd(&counts[8 + k], term[k]);
    if (lane == 0) __hip_atomic_fetch_add(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    sink[blockIdx.x * 512 + threadIdx.x] = y.x + y.y;
}

typedef void (*kern_t)(unsigned *, float *, int, int);
static void run(const char *what, kern_t kern, unsigned *d_counts, float *d_sink, int iters, int only_partner)
{
    for (int partner = 0; partner < 3; partner++) {
        if (only_partner >= 0 && partner != only_partner) continue;
        unsigned h[10];
        hipMemset(d_counts, 0, sizeof(h));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(2048), dim3(512), 0, 0, d_counts, d_sink, iters, partner);
        hipEventRecord(e1);
        hipError_t st = hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d_counts, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-58s partner %-5s %6.1f ms  wrong LOW halves by lane quarter %u %u %u %u (product gone %u, selection ignored %u)  wrong HIGH halves %u %u %u %u  of %.3g checks a quarter%s\n", what,
               partner == 0 ? "idle" : partner == 1 ? "mfma" : "valu", ms, h[0], h[1], h[2], h[3], h[8], h[9], h[4], h[5], h[6], h[7], 2048.0 * 4 * 16 * iters,
               st == hipSuccess ? "" : "  LAUNCH FAILED");
        fflush(stdout);
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned *d_counts; float *d_sink;
    hipMalloc(&d_counts, 128);
    hipMalloc(&d_sink, 2048 * 512 * 4);
    for (int partner = 0; partner < 2; partner++) {
        unsigned h[16];
        hipMemset(d_counts, 0, sizeof(h));
        hipLaunchKernelGGL(k_kernel_sequence, dim3(2048), dim3(512), 0, 0, d_counts, d_sink, iters, partner);
        hipDeviceSynchronize();
        hipMemcpy(h, d_counts, sizeof(h), hipMemcpyDeviceToHost);
        printf("the wide GEMM's epilogue sequence (weights on src0)          partner %-5s  wrong LOW halves by lane quarter %u %u %u %u  wrong HIGH halves %u %u %u %u  == the sum without term k: %u %u %u %u %u %u %u %u\n",
               partner ? "mfma" : "idle", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]);
        fflush(stdout);
    }
    run("v_pk_fma_f32 op_sel:[0,1,0], v_rcp_f32 pairs between", k_s1hi_rcp, d_counts, d_sink, iters, -1);
    run("v_pk_fma_f32 op_sel:[0,1,0], v_mul_f32 pairs between", k_s1hi_mul, d_counts, d_sink, iters, -1);
    run("v_pk_fma_f32 op_sel:[0,1,0], v_mov pairs + s_nop 1 between", k_s1hi_nop, d_counts, d_sink, iters, -1);
    run("v_pk_fma_f32 op_sel:[0,1,0], v_mov pairs + s_nop 7 between", k_s1hi_nop7, d_counts, d_sink, iters, 1);
    run("  the same (v_mul_f32 pairs) with s_nop 0 BEFORE it", k_s1hi_nopb, d_counts, d_sink, iters, 1);
    run("  the same (v_mul_f32 pairs) with s_nop 0 AFTER it", k_s1hi_nopa, d_counts, d_sink, iters, 1);
    run("v_pk_fma_f32 op_sel_hi:[1,0,1], v_mul_f32 pairs between", k_s1lo_mul, d_counts, d_sink, iters, 1);
    run("v_pk_fma_f32 (no selection), v_mul_f32 pairs between", k_none_mul, d_counts, d_sink, iters, 1);
    run("v_pk_fma_f32 op_sel:[1,0,0], v_mul_f32 pairs between", k_s0hi_mul, d_counts, d_sink, iters, 1);
    run("v_pk_fma_f32 op_sel:[0,0,1], v_mul_f32 pairs between", k_s2hi_mul, d_counts, d_sink, iters, 1);
    run("v_pk_fma_f32 op_sel:[1,0,0] (weights as src0), v_rcp_f32 pairs", k_w0hi_rcp, d_counts, d_sink, iters, 1);
    run("v_pk_fma_f32 op_sel:[1,0,0] (weights as src0), v_mul_f32 pairs", k_w0hi_mul, d_counts, d_sink, iters, 1);
    run("v_pk_fma_f32 op_sel:[1,0,0] (weights as src0), s_nop 1", k_w0hi_nop, d_counts, d_sink, iters, 1);
    run("v_pk_fma_f32 op_sel:[1,0,0] with D = S0, v_rcp_f32 pairs", k_dy_rcp, d_counts, d_sink, iters, 1);
    run("v_pk_fma_f32 op_sel:[1,0,0] with D = S0, s_nop 1", k_dy_nop, d_counts, d_sink, iters, 1);
    run("v_pk_fma_f32 op_sel_hi:[1,0,1], v_rcp_f32 pairs", k_s1lo_rcp, d_counts, d_sink, iters, 1);
    run("v_pk_fma_f32 op_sel_hi:[1,0,1], s_nop 1", k_s1lo_nop, d_counts, d_sink, iters, 1);
    run("v_pk_fma_f32 (no selection), s_nop 1", k_none_nop, d_counts, d_sink, iters, 1);
    run("v_pk_mul_f32 op_sel:[0,1] (+ v_pk_add_f32), v_mul between", k_mul_s1hi, d_counts, d_sink, iters, 1);
    run("v_pk_add_f32 op_sel:[0,1] (+ v_pk_mul_f32), v_mul between", k_add_s1hi, d_counts, d_sink, iters, 1);
    return 0;
}
