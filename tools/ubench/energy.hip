// energy.hip -- what an instruction costs in ENERGY on gfx950 (diagnostic, not part of the product).
//
//   ./energy <mode> <seconds>
//
// runs one loop kind back to back on every CU for <seconds> and prints its rate (wave-instructions per second over the
// chip), the clock the chip held inside the kernel (s_memtime over s_memrealtime) and the wall time per launch;
// tools/energy_probe.py samples the socket power beside it and divides.  Operands are random (zeros raise the clock:
// MI355X_MICROARCH.md, DVFS give-back).  Modes:
//   idle      every wave sleeps (s_sleep): the chip's floor with all CUs occupied
//   mfma16    v_mfma_f32_16x16x32_f16, A and C/D in accumulation registers, B architectural, 1 wave per SIMD
//   mfma16w2  the same with 2 waves per SIMD
//   mfma32    v_mfma_f32_32x32x16_f16, the same flops per iteration, 1 wave per SIMD
//   mfma32w2  2 waves per SIMD
//   fma       v_fma_f32 over 16 independent chains, 2 waves per SIMD
//   split     the f16 hi/lo split's mix: v_mul_f32 x2, v_cvt_pk_f16_f32, v_fma_mix_f32 x2, v_cvt_pk_f16_f32; 2 waves per SIMD
//   ldsr      ds_read_b128 streaming, 2 waves per SIMD
//   m16fma    mfma16w2's MFMAs with 3 v_fma_f32 behind each (what a fused block looks like)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int uint32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float rnd(unsigned s) { return (float)(hash(s) & 0xffffu) / 32768.0f - 1.0f; }   // [-1, 1)

enum { IDLE, MFMA16, MFMA32, FMA, SPLIT, LDSR, M16FMA, PKFMA, TRANS, DPPMOV };

template <int MODE>
__global__ void __launch_bounds__(512) k(int iters, unsigned long long *clk, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const unsigned seed = blockIdx.x * 1024u + threadIdx.x;
    uint32x4 *tab = reinterpret_cast<uint32x4 *>(smem);
    if (MODE == LDSR) {
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = uint32x4{hash(i), hash(i + 77777), hash(i + 1234567), hash(i * 3 + 1)};
        __syncthreads();
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
    if (MODE == IDLE) {
        for (int it = 0; it < iters; it++) __builtin_amdgcn_s_sleep(127);
    } else if (MODE == MFMA16 || MODE == M16FMA) {
        half8 A[4], B[2];
        for (int q = 0; q < 4; q++) for (int i = 0; i < 8; i++) A[q][i] = (_Float16)rnd(seed * 64 + q * 8 + i);
        for (int q = 0; q < 2; q++) for (int i = 0; i < 8; i++) B[q][i] = (_Float16)rnd(seed * 64 + 32 + q * 8 + i);
        floatx4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        float f[12];
        for (int i = 0; i < 12; i++) f[i] = rnd(seed * 64 + 50 + i);
        const float m = -0.99993f, c = 0.37f * rnd(seed);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 8; q++) {       // 8 MFMAs of 16 cycles
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[q & 3]) : "a"(A[q & 3]), "v"(B[q & 1]));
                if (MODE == M16FMA) {
#pragma unroll
                    for (int r = 0; r < 3; r++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[(q * 3 + r) % 12]) : "v"(m), "v"(c));
                }
            }
        }
        for (int q = 0; q < 4; q++) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
        for (int i = 0; i < 12; i++) s += f[i];
    } else if (MODE == MFMA32) {
        half8 A[4], B[2];
        for (int q = 0; q < 4; q++) for (int i = 0; i < 8; i++) A[q][i] = (_Float16)rnd(seed * 64 + q * 8 + i);
        for (int q = 0; q < 2; q++) for (int i = 0; i < 8; i++) B[q][i] = (_Float16)rnd(seed * 64 + 32 + q * 8 + i);
        floatx16 acc[2] = {{0}, {0}};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 4; q++)         // 4 MFMAs of 32 cycles: the same flops as 8 of the other shape
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[q & 1]) : "a"(A[q & 3]), "v"(B[q & 1]));
        }
        for (int q = 0; q < 2; q++) for (int i = 0; i < 16; i++) s += acc[q][i];
    } else if (MODE == FMA) {
        float f[16];
        for (int i = 0; i < 16; i++) f[i] = rnd(seed * 64 + i);
        const float m = -0.99993f, c = 0.37f * rnd(seed);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 32; q++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[q & 15]) : "v"(m), "v"(c));
        }
        for (int i = 0; i < 16; i++) s += f[i];
    } else if (MODE == PKFMA) {                       // packed fp32: two multiply-adds a lane and instruction
        floatx2 f[16];
        for (int i = 0; i < 16; i++) f[i] = floatx2{rnd(seed * 64 + i), rnd(seed * 64 + 32 + i)};
        const floatx2 m = {-0.99993f, -0.99991f}, c = {0.37f * rnd(seed), 0.21f * rnd(seed + 9)};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 32; q++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(f[q & 15]) : "v"(m), "v"(c));
        }
        for (int i = 0; i < 16; i++) s += f[i][0] + f[i][1];
    } else if (MODE == TRANS) {                       // the transcendental unit: exp2 and rcp alternating (the wide engine's epilogue pair)
        float f[16];
        for (int i = 0; i < 16; i++) f[i] = rnd(seed * 64 + i);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 32; q++) {
                if (q & 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[q & 15]));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(f[q & 15]));
            }
        }
        for (int i = 0; i < 16; i++) s += f[i];
    } else if (MODE == DPPMOV) {                      // a DPP row shift into a fresh register (the block-transform kernel's sliding sums)
        float f[16];
        for (int i = 0; i < 16; i++) f[i] = rnd(seed * 64 + i);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 32; q++) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(f[q & 15]) : "v"(f[(q + 5) & 15]));
        }
        for (int i = 0; i < 16; i++) s += f[i];
    } else if (MODE == SPLIT) {
        float x[8];
        for (int i = 0; i < 8; i++) x[i] = rnd(seed * 64 + i);
        const float sx = 8192.0f;
        unsigned accu = 0;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {       // 4 pairs: 6 instructions each = 24 per iteration
                float ta, tb, ra, rb;
                unsigned h, l;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ta) : "v"(x[2 * q]), "v"(sx));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(tb) : "v"(x[2 * q + 1]), "v"(sx));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(ta), "v"(tb));
                asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(x[2 * q]), "v"(sx), "v"(h));
                asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(x[2 * q + 1]), "v"(sx), "v"(h));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(ra), "v"(rb));
                accu ^= h ^ l;
                x[2 * q] = -x[2 * q];
            }
        }
        s = (float)accu + x[0];
    } else if (MODE == LDSR) {
        uint32x4 a = {0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                uint32x4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)((((it * 16 + q) * 64 + lane) & 4095) * 16)));
                asm volatile("s_waitcnt lgkmcnt(4)");
                a ^= v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
        s = (float)(a[0] ^ a[1] ^ a[2] ^ a[3]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        clk[2 * w] = t1 - t0;
        clk[2 * w + 1] = r1 - r0;
    }
    if (s == 12345.678f) sink[0] = s;
}

// hbm / hbmnt: every workgroup sums a contiguous run of a 4 GiB buffer of random bits, 8 x 16-byte loads in flight per thread
// (plain or non-temporal): what moving a byte from HBM into a CU costs
typedef float floatx4r __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) fill(uint32x4 *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned a = hash((unsigned)i), b = hash((unsigned)i + 0x9e3779b9u);
        p[i] = uint32x4{(a & 0x807fffffu) | 0x3c000000u, (b & 0x807fffffu) | 0x3c000000u, ((a >> 3) & 0x807fffffu) | 0x3c000000u, ((b >> 5) & 0x807fffffu) | 0x3c000000u};
    }
}
template <bool NT>
__global__ void __launch_bounds__(256) rd(const floatx4r *__restrict__ p, size_t n, size_t per_wg, float *sink)
{
    floatx4r acc = {0, 0, 0, 0};
    const size_t b = (size_t)blockIdx.x * per_wg, e = b + per_wg < n ? b + per_wg : n;
    for (size_t i = b + threadIdx.x; i < e; i += 256 * 8) {
        floatx4r v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = i + 256 * u < e ? (NT ? __builtin_nontemporal_load(p + i + 256 * u) : p[i + 256 * u]) : floatx4r{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
// the same read through a buffer descriptor with a cache-policy operand: aux bit 0 sc0, bit 1 nt, bit 4 sc1
template <int AUX>
__global__ void __launch_bounds__(256) rdaux(const floatx4r *__restrict__ p, size_t n, size_t per_wg, float *sink)
{
    floatx4r acc = {0, 0, 0, 0};
    const size_t b = (size_t)blockIdx.x * per_wg, e = b + per_wg < n ? b + per_wg : n;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<floatx4r *>(p + b), 0, (int)((e - b) * 16), 0x00020000);
    for (size_t i = threadIdx.x; i < e - b; i += 256 * 8) {
        uint32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((i + 256 * u) * 16), 0, AUX);
#pragma unroll
        for (int u = 0; u < 8; u++) { acc[0] += __uint_as_float(v[u][0]); acc[1] += __uint_as_float(v[u][1]); acc[2] += __uint_as_float(v[u][2]); acc[3] += __uint_as_float(v[u][3]); }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
template <int AUX>
void run_hbm_aux(const char *name, double secs)
{
    const size_t n = (size_t)1 << 28;
    floatx4r *p; float *sink;
    CK(hipMalloc(&p, n * 16));
    CK(hipMalloc(&sink, 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32x4 *>(p), n);
    CK(hipDeviceSynchronize());
    const int grid = 2048;
    const size_t per_wg = (n + grid - 1) / grid;
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    while (el < secs) {
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(rdaux<AUX>, dim3(grid), dim3(256), 0, 0, p, n, per_wg, sink);
        CK(hipDeviceSynchronize());
        launches += 50;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("%s: %.3f ms per 4 GiB launch = %.2f TB/s, %.4e wave-instructions/s over the chip (1 KiB each)\n", name, el / launches * 1e3,
           (double)launches * n * 16 / el / 1e12, (double)launches * n * 16 / 1024 / el);
    CK(hipFree(p)); CK(hipFree(sink));
}

template <bool NT>
void run_hbm(const char *name, double secs)
{
    const size_t n = (size_t)1 << 28;
    floatx4r *p; float *sink;
    CK(hipMalloc(&p, n * 16));
    CK(hipMalloc(&sink, 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32x4 *>(p), n);
    CK(hipDeviceSynchronize());
    const int grid = 2048;
    const size_t per_wg = (n + grid - 1) / grid;
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    while (el < secs) {
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(rd<NT>, dim3(grid), dim3(256), 0, 0, p, n, per_wg, sink);
        CK(hipDeviceSynchronize());
        launches += 50;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("%s: %.3f ms per 4 GiB launch = %.2f TB/s, %.4e wave-instructions/s over the chip (1 KiB each)\n", name, el / launches * 1e3,
           (double)launches * n * 16 / el / 1e12, (double)launches * n * 16 / 1024 / el);
    CK(hipFree(p)); CK(hipFree(sink));
}

template <int MODE>
void run(const char *name, int waves, double secs, int per_iter, int iters)
{
    const int blocks = 256;
    unsigned long long *clk; float *sink;
    CK(hipMalloc(&clk, blocks * 8 * 2 * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, 4));
    CK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));    // one workgroup per CU
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves), 100 * 1024, 0, iters, clk, sink);
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    while (el < secs) {
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves), 100 * 1024, 0, iters, clk, sink);
        CK(hipDeviceSynchronize());
        launches += 20;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    std::vector<unsigned long long> h(blocks * waves * 2);
    CK(hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int w = 0; w < blocks * waves; w++) { cyc += (double)h[2 * w]; real += (double)h[2 * w + 1]; }
    const double ghz = cyc / real * 0.1, wave_instr = (double)launches * blocks * waves * (double)iters * per_iter;
    printf("%s: waves/SIMD %d, %.3f ms per launch, in-kernel clock %.3f GHz, %.4e wave-instructions/s over the chip, %.2f clocks per instruction and SIMD\n",
           name, waves / 4, el / launches * 1e3, ghz, wave_instr / el, (cyc / (blocks * waves)) / ((double)iters * per_iter) * 1.0 / (waves / 4));
    CK(hipFree(clk)); CK(hipFree(sink));
}

int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "mfma16";
    const double secs = argc > 2 ? atof(argv[2]) : 5.0;
    if (!strcmp(mode, "idle")) run<IDLE>(mode, 8, secs, 1, 2000);
    else if (!strcmp(mode, "mfma16")) run<MFMA16>(mode, 4, secs, 8, 60000);
    else if (!strcmp(mode, "mfma16w2")) run<MFMA16>(mode, 8, secs, 8, 30000);
    else if (!strcmp(mode, "mfma32")) run<MFMA32>(mode, 4, secs, 4, 60000);
    else if (!strcmp(mode, "mfma32w2")) run<MFMA32>(mode, 8, secs, 4, 30000);
    else if (!strcmp(mode, "fma")) run<FMA>(mode, 8, secs, 32, 60000);
    else if (!strcmp(mode, "split")) run<SPLIT>(mode, 8, secs, 24, 60000);
    else if (!strcmp(mode, "ldsr")) run<LDSR>(mode, 8, secs, 16, 60000);
    else if (!strcmp(mode, "hbm")) run_hbm<false>(mode, secs);
    else if (!strcmp(mode, "hbmnt")) run_hbm<true>(mode, secs);
    else if (!strcmp(mode, "aux0")) run_hbm_aux<0>(mode, secs);
    else if (!strcmp(mode, "aux1")) run_hbm_aux<1>(mode, secs);
    else if (!strcmp(mode, "aux2")) run_hbm_aux<2>(mode, secs);
    else if (!strcmp(mode, "aux3")) run_hbm_aux<3>(mode, secs);
    else if (!strcmp(mode, "aux16")) run_hbm_aux<16>(mode, secs);
    else if (!strcmp(mode, "aux17")) run_hbm_aux<17>(mode, secs);
    else if (!strcmp(mode, "aux18")) run_hbm_aux<18>(mode, secs);
    else if (!strcmp(mode, "aux19")) run_hbm_aux<19>(mode, secs);
    else if (!strcmp(mode, "m16fma")) run<M16FMA>(mode, 8, secs, 8, 30000);
    else if (!strcmp(mode, "pkfma")) run<PKFMA>(mode, 8, secs, 32, 60000);
    else if (!strcmp(mode, "trans")) run<TRANS>(mode, 8, secs, 32, 30000);
    else if (!strcmp(mode, "dppmov")) run<DPPMOV>(mode, 8, secs, 32, 60000);
    else { printf("unknown mode %s\n", mode); return 1; }
    return 0;
}
