// kernels_fixup.hip -- exact recomputation of the evaluations (or spectrogram frames) the fused kernels reported through
// their precision guard (kernels.hpp, FixItem): windows too close to the floor of the block-floating-point grid they were
// computed on, passes with an infinite sample, level steps of hundreds of dB.
//
// The same path, from the samples, with no grid at all (reference, root relative):
//   extractPower          Common/CircularShortTimeFourierTransform.swift:280-337   -- here by the DFT definition in fp64 over
//                                                                                     the fp32 samples and the fp32 window table
//   processFourierData    Common/SyllableDetector.swift:134-151
//   processNewValue       Common/SyllableDetector.swift:153-217
//   NeuralNet.apply       Common/NeuralNet.swift:294-326, :366-377                 -- unfolded, the reference's operation order
//   lastDetected          Common/SyllableDetector.swift:27-31
// so that NaN appears exactly in the evaluations whose windows contain the offending sample, as in the reference.
//
// Launched behind the fused kernel on its stream with a fixed grid: a workgroup walks the work list from its own index in
// steps of the grid size; an empty list costs one load.  Rare by construction -- this is the slow, careful path.
//
// gfx950 only: wave = 64 lanes, 256-thread workgroups.

#include "generic_eval.hpp"

namespace sd {

namespace {

using namespace generic_dev;

constexpr int kBlock = 256;
constexpr int kFixGrid = 512;                   // two workgroups a CU when the list is long; an empty list -- the ordinary case -- still costs one launch of a few microseconds (MEASUREMENTS R5.7)
constexpr int kMaxFrames = kFixMaxCount + 11;      // frames behind one item: its evaluations' windows (timeRange <= 12)

__global__ void __launch_bounds__(kBlock)
fixup_kernel(const FixDesc fd, const NetDesc n, const float *__restrict__ samples, int64_t stride, int64_t J, int64_t E,
             float *__restrict__ outputs, uint8_t *__restrict__ flags, float *__restrict__ columns, const FixList list)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid / kWave, lane = tid & (kWave - 1);
    const int N = fd.N, W = fd.W, F = fd.F, T = fd.T;
    // LDS: trigonometric table [N] double2 | samples of the item [span] | window [W] | columns [kMaxFrames][F] | network buffers
    double2 *ctab = reinterpret_cast<double2 *>(smem);
    const int span_max = (kMaxFrames - 1) * fd.hop + W;
    float *xs = reinterpret_cast<float *>(ctab + N);
    float *win = xs + span_max;
    float *cols = win + W;
    float *nbuf = cols + kMaxFrames * F;

    unsigned count = list.counters[0];
    count = count < list.capacity ? count : list.capacity;
    if (count == 0) {
        // the ordinary case, behind every batch call: nothing to reset (the counters are as a finished launch leaves them), so no
        // workgroup takes part in the last-one-out protocol below -- 512 atomics on one word are 10 us, this is a bare launch
        if (blockIdx.x == 0 && tid == 0) {
            list.counters[2] = 0u;
            if (list.host_count) *list.host_count = 0u;
        }
        return;
    }
    // (the constant 100 MHz counter; stored and read back at device scope: the reader's CU may hold the line from its read of the count)
    if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(list.counters + 4, (unsigned)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
        for (int i = tid; i < N; i += kBlock) ctab[i] = fd.ctab[i];
        for (int i = tid; i < W; i += kBlock) win[i] = fd.window[i];
    }
    for (unsigned it = blockIdx.x; it < count; it += gridDim.x) {
        const FixItem item = list.items[it];
        const int c = item.c;
        const bool spect = item.kind != 0;
        const int64_t first = (int64_t)item.first;
        int nframes = spect ? item.count : item.count + T - 1;
        if (first + nframes > J) nframes = (int)(J - first);
        if (nframes > kMaxFrames) nframes = kMaxFrames;       // (never: items hold at most kFixMaxCount)
        const float *row = samples + (int64_t)c * stride + first * fd.hop + fd.gap;
        const int span = nframes > 0 ? (nframes - 1) * fd.hop + W : 0;
        __syncthreads();                                      // the previous item's columns and samples are no longer read
        for (int i = tid; i < span; i += kBlock) xs[i] = row[i];
        __syncthreads();
        // |X[k]| for every (frame, bin) of the item: X[k] = sum_n x[n] w[n] e^{-2 pi i k n / N} (:311-333; bin 0 is real, the
        // packed Nyquist term is dropped :323 -- bins stay below N/2), one task per thread at a time, the angle index k n mod N
        // advanced incrementally (exact)
        for (int task = tid; task < nframes * F; task += kBlock) {
            const int fr = task / F, b = task - fr * F, k = fd.f0 + b;
            const float *x = xs + fr * fd.hop;
            double re = 0.0, im = 0.0;
            int idx = 0;
            for (int i = 0; i < W; i++) {
                const double xw = (double)x[i] * (double)win[i];
                const double2 cs = ctab[idx];
                re = fma(xw, cs.x, re);
                im = fma(-xw, cs.y, im);
                idx = (idx + k) & (N - 1);
            }
            const float mag = (float)sqrt(re * re + im * im);
            cols[fr * F + b] = mag;
            if (spect && columns)                                 // zvabs / 2 :329-333, or zvmags / 4 :270-274
                columns[((int64_t)c * J + first + fr) * F + b] = fd.power_mode ? (float)(re * re + im * im) : mag;
        }
        __syncthreads();
        if (spect || (!outputs && !flags)) continue;    // (no network buffers were allocated for a launch without result arrays)
        // the item's evaluations, one wave each a round; every wave makes the same number of rounds (workgroup barriers inside)
        float *bufA = nbuf + (size_t)wave * 2 * n.max_width, *bufB = bufA + n.max_width;
        for (int r = 0; r < (kFixMaxCount + kBlock / kWave - 1) / (kBlock / kWave); r++) {
            const int el = r * (kBlock / kWave) + wave;
            const int64_t e = first + el;
            const bool valid = el < item.count && e < E;
            mlp_eval_wave(n, cols + (valid ? el : 0) * F, valid, bufA, bufB, lane,
                          (valid && outputs) ? outputs + ((int64_t)c * E + e) * n.n_out : nullptr,
                          (valid && flags) ? flags + (int64_t)c * E + e : nullptr);
        }
    }
    // the last workgroup out resets the list for the next launch (every workgroup has read the count by then) and keeps the
    // number of items for syldet_fixup_stats
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        const unsigned done = atomicAdd(list.counters + 1, 1u);
        if (done == gridDim.x - 1) {
            if (list.host_count) {                    // the launch times itself: no events of its own around a launch that is empty on ordinary audio
                list.host_count[1] = (unsigned)__builtin_amdgcn_s_memrealtime() - __hip_atomic_load(list.counters + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                list.host_count[0] = list.counters[0];
            }
            list.counters[2] = list.counters[0];
            list.counters[0] = 0u;
            list.counters[1] = 0u;
            __threadfence();
        }
    }
}

}  // namespace

hipError_t launch_fixup(const FixDesc &fd, const NetDesc &n, const float *samples, int64_t stride, int64_t J, int64_t E,
                        float *outputs, uint8_t *flags, float *columns, const FixList &list, hipStream_t stream)
{
    if (!list.counters) return hipSuccess;
    // (the network's buffers only where evaluations are recomputed: behind the spectrogram kernels -- outputs and flags null,
    // frame items only -- a wide network in front of the generic / wide engines must not count against the 160 KB)
    const bool evals = outputs != nullptr || flags != nullptr;
    const size_t lds = (size_t)fd.N * sizeof(double2) + ((size_t)(kMaxFrames - 1) * fd.hop + 2 * (size_t)fd.W + (size_t)kMaxFrames * fd.F) * sizeof(float) +
                       (evals ? (size_t)(kBlock / kWave) * 2 * (size_t)n.max_width * sizeof(float) : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t st = hipFuncSetAttribute((const void *)fixup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (st != hipSuccess) return st;
    }
    hipLaunchKernelGGL(fixup_kernel, dim3(kFixGrid), dim3(kBlock), lds, stream, fd, n, samples, stride, J, E, outputs, flags, columns, list);
    return hipGetLastError();
}

}  // namespace sd
