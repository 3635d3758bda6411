// What a read-only kernel reaches on this part (diagnostic, not part of the product; cited in DESIGN.md §6.0): every thread
// sums 16-byte loads of a 4 GiB buffer, U loads in flight, grid-stride or one contiguous run per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/read_bw tools/ubench/read_bw.hip && gpurun -- ./tools/ubench/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int U, bool CONTIG>
__global__ void __launch_bounds__(256) rd(const floatx4 *__restrict__ p, size_t n, size_t per_wg, float *sink)
{
    floatx4 acc = {0, 0, 0, 0};
    if (CONTIG) {                       // one contiguous run per workgroup (the fused kernel's shape: a segment of a channel)
        const size_t b = (size_t)blockIdx.x * per_wg, e = b + per_wg < n ? b + per_wg : n;
        for (size_t i = b + threadIdx.x; i < e; i += 256 * U) {
            floatx4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = i + 256 * u < e ? __builtin_nontemporal_load(p + i + 256 * u) : floatx4{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < U; u++) acc += v[u];
        }
    } else {
        const size_t stride = (size_t)gridDim.x * 256 * U;
        for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
            floatx4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = i + 256 * u < n ? __builtin_nontemporal_load(p + i + 256 * u) : floatx4{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < U; u++) acc += v[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int U, bool CONTIG>
void run(const char *name, const floatx4 *p, size_t n, int grid, float *sink)
{
    const size_t per_wg = (n + grid - 1) / grid;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((rd<U, CONTIG>), dim3(grid), dim3(256), 0, 0, p, n, per_wg, sink);
    CK(hipEventRecord(a));
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((rd<U, CONTIG>), dim3(grid), dim3(256), 0, 0, p, n, per_wg, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s grid %6d  %.3f ms  %.2f TB/s\n", name, grid, ms / 10, n * 16.0 / (ms / 10 * 1e-3) / 1e12);
    fflush(stdout);
}

int main()
{
    const size_t n = (size_t)1 << 28;          // 4 GiB of 16-byte elements
    floatx4 *p; float *sink;
    CK(hipMalloc(&p, n * 16));
    CK(hipMemset(p, 0, n * 16));
    CK(hipMalloc(&sink, 4));
    for (int grid : {256, 512, 1024, 2048, 8192}) {
        run<4, false>("grid-stride, 4 loads in flight", p, n, grid, sink);
        run<8, false>("grid-stride, 8 loads in flight", p, n, grid, sink);
        run<16, false>("grid-stride, 16 loads in flight", p, n, grid, sink);
        run<8, true>("contiguous run per workgroup, 8 in flight", p, n, grid, sink);
        run<16, true>("contiguous run per workgroup, 16 in flight", p, n, grid, sink);
    }
    return 0;
}
