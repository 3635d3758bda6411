// Calibration microbenchmarks for gfx950 (diagnostic, not part of the product):
//   mfma:   back-to-back v_mfma_f32_16x16x32_f16 with 4 independent accumulators, W waves per CU
//   lds:    ds_read_b128 streaming from a 64 KB table, W waves per CU
//   both:   the fused kernel's k-step shape: 8 b128 + 4 b64 reads, 12 MFMAs
// prints cycles (s_memtime) and wall-clock per inner iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(512) k(int iters, unsigned long long *cyc, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32x4 *tab = reinterpret_cast<uint32x4 *>(smem);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = uint32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    floatx4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    uint32x4 a[8];
    for (int i = 0; i < 8; i++) a[i] = tab[i * 64 + lane];
    uint32x4 b0 = tab[lane], b1 = tab[64 + lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        union { uint32x4 u; half8 h; } A[8], B0, B1;
        for (int i = 0; i < 8; i++) A[i].u = a[i];
        B0.u = b0; B1.u = b1;
        if (MODE != 0) {   // lds or both: fetch the next fragments
            const int ks = (it + 1) & 7;
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = tab[(ks * 8 + i) * 64 + lane];
            b0 = tab[(ks * 8) * 64 + ((lane * 5) & 63)];
            b1 = tab[(ks * 8 + 1) * 64 + ((lane * 5) & 63)];
        }
        if (MODE != 1) {   // mfma or both
#pragma unroll
            for (int m = 0; m < 4; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[2 * m].h, B0.h, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[2 * m].h, B1.h, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[2 * m + 1].h, B0.h, acc[m], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("" ::"v"(A[i].u));
            asm volatile("" ::"v"(B0.u), "v"(B1.u));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    float s = 0;
    for (int m = 0; m < 4; m++) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    for (int i = 0; i < 8; i++) s += (float)a[i][0];
    if (s == 12345.678f) sink[0] = s;
}

// MODE 3: the same work with v_mfma_f32_32x32x16_f16: per iteration 8 b128 fetches (two k-steps of 16: A hi, A lo, B hi, B lo each)
// and 6 MFMAs of 32 cycles = the 192 matrix cycles of MODE 2's 12 x 16, with 8 KB instead of 10 KB from LDS
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(512) k32(int iters, unsigned long long *cyc, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32x4 *tab = reinterpret_cast<uint32x4 *>(smem);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = uint32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    floatx16 acc = {0};
    uint32x4 a[8];
    for (int i = 0; i < 8; i++) a[i] = tab[i * 64 + lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        union { uint32x4 u; half8 h; } A[8];
        for (int i = 0; i < 8; i++) A[i].u = a[i];
        const int ks = (it + 1) & 7;
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = tab[(ks * 8 + i) * 64 + ((lane * (i & 2 ? 5 : 1)) & 63)];
        // two k-steps: (Ahi, Alo, Bhi, Blo) = A[0..3], A[4..7]
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0].h, A[2].h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0].h, A[3].h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1].h, A[2].h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[4].h, A[6].h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[4].h, A[7].h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[5].h, A[6].h, acc, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    float s = 0;
    for (int m = 0; m < 16; m++) s += acc[m];
    for (int i = 0; i < 8; i++) s += (float)a[i][0];
    if (s == 12345.678f) sink[0] = s;
}

// coissue: do matrix and vector instructions overlap on one SIMD?  Per iteration NM MFMAs (4 independent
// accumulators) and NV v_fma_f32 (16 independent chains).  ROLE 0: every wave issues both (same wave);
// ROLE 1: waves 0-3 (one per SIMD) issue the MFMAs, waves 4-7 the FMAs; NM or NV = 0 gives each alone.
template <int NM, int NV, int ROLE>
__global__ void __launch_bounds__(512) kco(int iters, unsigned long long *cyc, float *sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    floatx4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    half8 A, B;
    for (int i = 0; i < 8; i++) { A[i] = (_Float16)(lane * 0.001f); B[i] = (_Float16)(i * 0.01f); }
    float f[16];
    for (int i = 0; i < 16; i++) f[i] = lane + i;
    const float m = 1.0f + 1e-7f * lane, c = 1e-3f;
    const bool do_m = ROLE == 0 || wave < 4, do_v = ROLE == 0 || wave >= 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (do_m) {
#pragma unroll
            for (int q = 0; q < NM; q++) acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc[q & 3], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int q = 0; q < NV; q++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[q & 15]) : "v"(m), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
    float s = 0;
    for (int q = 0; q < 4; q++) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    for (int i = 0; i < 16; i++) s += f[i];
    if (s == 12345.678f) sink[0] = s;
}

template <int NM, int NV, int ROLE>
void run_co(const char *name, int waves, int iters)
{
    unsigned long long *cyc; float *sink;
    const int blocks = 256;
    CK(hipMalloc(&cyc, blocks * 16 * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)kco<NM, NV, ROLE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((kco<NM, NV, ROLE>), dim3(blocks), dim3(64 * waves), 150 * 1024, 0, iters, cyc, sink);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s waves/CU=%2d  MFMA/iter=%2d FMA/iter=%2d   wall ns/iter=%8.2f\n", name, waves, NM, NV, ms * 1e6 / iters);
    CK(hipFree(cyc)); CK(hipFree(sink));
}


// interleave: one wave per SIMD, every instruction an ordered assembly statement: per iteration 8 x [one MFMA, NV
// independent v_fma_f32].  ACC 0: C/D in architectural registers and A in accumulation registers (the arrangement of
// kernels_fused_r.hip's first version); ACC 1: C/D in accumulation registers, A and B architectural (what the compiler
// picks by itself); ACC 2: C/D and A in accumulation registers, B architectural.
template <int NV, int ACC>
__global__ void __launch_bounds__(256, 1) kil(int iters, unsigned long long *cyc, float *sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    floatx4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    half8 A, B;
    for (int i = 0; i < 8; i++) { A[i] = (_Float16)(lane * 0.001f); B[i] = (_Float16)(i * 0.01f); }
    float f[16];
    for (int i = 0; i < 16; i++) f[i] = lane + i;
    const float m = 1.0f + 1e-7f * lane, c = 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (ACC == 0 || ACC == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[q & 3]) : "a"(A), "v"(B));
            else if (ACC == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[q & 3]) : "v"(A), "v"(B));
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[q & 3]) : "a"(A), "v"(B));
#pragma unroll
            for (int r = 0; r < NV; r++) {
                if (ACC < 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[(q * NV + r) & 15]) : "v"(m), "v"(c));
                else asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(f[(q * NV + r) & 15]) : "v"(m), "v"(c));   // ACC 3: ACC 0's MFMA, the staging's split instruction
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
    float s = 0;
    for (int q = 0; q < 4; q++) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    for (int i = 0; i < 16; i++) s += f[i];
    if (s == 12345.678f) sink[0] = s;
}

template <int NV, int ACC>
void run_il(int iters)
{
    unsigned long long *cyc; float *sink;
    const int blocks = 256;
    CK(hipMalloc(&cyc, blocks * 16 * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)kil<NV, ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((kil<NV, ACC>), dim3(blocks), dim3(256), 150 * 1024, 0, iters, cyc, sink);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("interleaved, 1 wave/SIMD, %s: 8 x [MFMA + %d FMA]   wall ns/iter=%8.2f  = %6.2f ns per MFMA group\n",
           ACC == 3 ? "C/D arch, A acc, v_fma_mixlo_f16" : ACC == 0 ? "C/D arch, A acc " : ACC == 1 ? "C/D acc, A arch " : "C/D acc, A acc  ", NV, ms * 1e6 / iters, ms * 1e6 / iters / 8);
    CK(hipFree(cyc)); CK(hipFree(sink));
}

template <int MODE>
void run(const char *name, int waves, int iters)
{
    unsigned long long *cyc; float *sink;
    const int blocks = 256;
    CK(hipMalloc(&cyc, blocks * 16 * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k<(MODE == 3 ? 2 : MODE)>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CK(hipFuncSetAttribute((const void *)k32, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        if (MODE == 3) hipLaunchKernelGGL(k32, dim3(blocks), dim3(64 * waves), 150 * 1024, 0, iters, cyc, sink);
        else hipLaunchKernelGGL(k<(MODE == 3 ? 2 : MODE)>, dim3(blocks), dim3(64 * waves), 150 * 1024, 0, iters, cyc, sink);   // 150 KB LDS: one block per CU
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * waves);
    CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
    printf("%-5s waves/CU=%2d  memtime ticks/iter=%8.1f   wall ns/iter=%8.2f   (ticks/us=%.1f)\n", name, waves, avg / iters, ms * 1e6 / iters, avg / (ms * 1e3));
    CK(hipFree(cyc)); CK(hipFree(sink));
}

int main()
{
    const int iters = 20000;
    run_il<0, 0>(iters); run_il<1, 0>(iters); run_il<2, 0>(iters); run_il<3, 0>(iters); run_il<4, 0>(iters); run_il<6, 0>(iters);
    run_il<1, 3>(iters); run_il<2, 3>(iters); run_il<3, 3>(iters); run_il<4, 3>(iters); run_il<6, 3>(iters);
    run_il<0, 2>(iters); run_il<1, 2>(iters); run_il<2, 2>(iters); run_il<3, 2>(iters); run_il<4, 2>(iters);
    run_il<5, 2>(iters); run_il<6, 2>(iters); run_il<7, 2>(iters); run_il<8, 2>(iters); run_il<9, 2>(iters); run_il<10, 2>(iters); run_il<12, 2>(iters); run_il<14, 2>(iters);
    run_il<5, 0>(iters); run_il<7, 0>(iters); run_il<8, 0>(iters); run_il<10, 0>(iters);
    run_il<0, 1>(iters); run_il<1, 1>(iters); run_il<2, 1>(iters); run_il<3, 1>(iters); run_il<4, 1>(iters); run_il<6, 1>(iters);
    // one wave per SIMD: MFMA alone, FMA alone, both in one wave
    run_co<8, 0, 0>("mfma alone (1 wave/SIMD)", 4, iters);
    run_co<0, 24, 0>("fma alone (1 wave/SIMD)", 4, iters);
    run_co<8, 24, 0>("both, same wave (1 wave/SIMD)", 4, iters);
    run_co<8, 8, 0>("both, same wave (1 wave/SIMD)", 4, iters);
    // two waves per SIMD: each alone, both in every wave, and split by role
    run_co<8, 0, 0>("mfma alone (2 waves/SIMD)", 8, iters);
    run_co<0, 24, 0>("fma alone (2 waves/SIMD)", 8, iters);
    run_co<8, 24, 0>("both, same wave (2 waves/SIMD)", 8, iters);
    run_co<8, 24, 1>("mfma waves + fma waves (1+1/SIMD)", 8, iters);
    run_co<8, 48, 1>("mfma waves + fma waves (1+1/SIMD)", 8, iters);
    for (int w : {4, 8}) run<0>("mfma", w, iters);
    for (int w : {4, 8}) run<1>("lds", w, iters);
    for (int w : {4, 8}) run<2>("both", w, iters);
    for (int w : {4, 8}) run<3>("b32", w, iters);
    return 0;
}
