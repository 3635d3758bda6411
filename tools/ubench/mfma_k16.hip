// mfma_k16.hip -- clocks per v_mfma_f32_16x16x16_bf16 against v_mfma_f32_16x16x32_bf16 on gfx950 (is a K tail of 16 half the price?)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_k16 tools/ubench/mfma_k16.hip && gpurun -- ./tools/ubench/mfma_k16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short shortx4 __attribute__((ext_vector_type(4)));
template <int K32>
__global__ void __launch_bounds__(256) k(unsigned long long *out, float *sink, int iters)
{
    floatx4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a8, b8;
    shortx4 a4, b4;
    for (int i = 0; i < 8; i++) { a8[i] = (__bf16)(threadIdx.x * 0.001f + i); b8[i] = (__bf16)(i * 0.5f); }
    for (int i = 0; i < 4; i++) { a4[i] = (short)(threadIdx.x + i); b4[i] = (short)(i * 3); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            if (K32) acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[q & 3], 0, 0, 0);
            else acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[q & 3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
int main()
{
    unsigned long long *d; float *s;
    hipMalloc(&d, 8); hipMalloc(&s, 1024 * 256 * 4);
    const int iters = 20000;
    for (int rep = 0; rep < 2; rep++) {
        unsigned long long h;
        hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, d, s, iters); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("v_mfma_f32_16x16x32_bf16: %.2f clocks each (1 wave / SIMD, 4 accumulators)\n", (double)h / (16.0 * iters));
        hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, d, s, iters); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("v_mfma_f32_16x16x16_bf16: %.2f clocks each\n", (double)h / (16.0 * iters));
    }
    return 0;
}
