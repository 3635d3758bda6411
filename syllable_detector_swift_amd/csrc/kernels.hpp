// kernels.hpp -- device-side descriptors and launcher prototypes shared by the engine
// files.  Everything here is gfx950-only HIP.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace sd {

constexpr int kMaxFns = 8;
constexpr int kMaxLayers = 8;

// STFT geometry + tables (built once per handle by syldet_create).
struct StftDesc {
    int N, W, M, logM;        // fourierLength, windowLength, N/2, log2(N/2)
    int hop, gap;             // frame j covers samples [j*hop + gap, j*hop + gap + W)
    int f0, F;                // band [f0, f0+F)
    int power_mode;           // 0: |X| (extractPower), 1: |X|^2 (extractMagnitude)
    const float *window;      // [W]
    const float2 *tw;         // [M/2]  e^{-2 pi i t / M}     (complex FFT stage twiddles)
    const float2 *sw;         // [M]    e^{-2 pi i k / N}     (real-split twiddles)
};

struct DevFn {
    int kind;                 // syldet_fn_kind_t
    int xoff, gain;           // offsets (in floats) into the parameter blob
    float y;
};
struct DevLayer {
    int in, out, tf;          // tf: syldet_transfer_t
    int w, b;                 // offsets into the parameter blob; w is row-major [out][in]
};
// The network exactly as configured (generic engine: no algebraic folding).
struct NetDesc {
    int n_in_fns, n_layers, n_out_fns;
    DevFn in_fns[kMaxFns];
    DevLayer layers[kMaxLayers];
    DevFn out_fns[kMaxFns];
    int I, n_out, max_width;
    int scaling;              // syldet_scaling_t
    int rule;                 // syldet_rule_t
    const float *params;      // parameter blob
    const double *thresholds; // [n_out]
};

// ---- generic engine (any power-of-two N, any processing chain, any layer sizes) ----
// columns [C][J][F] <- samples [C][stride]
hipError_t launch_stft_generic(const StftDesc &d, const float *samples, int64_t stride, int C, int64_t J,
                               float *columns, hipStream_t stream);
// outputs [C][E][n_out], flags [C][E] <- columns [C][J][F]; either output may be null
hipError_t launch_mlp_generic(const NetDesc &n, int F, const float *columns, int C, int64_t J, int64_t E,
                              float *outputs, uint8_t *flags, hipStream_t stream);
// indices [C][capacity], counts [C] <- flags [C][E]
hipError_t launch_detections(const uint8_t *flags, int C, int64_t E, int64_t first_index, int64_t hop,
                             int64_t debounce_frames, int64_t *indices, int64_t capacity, int64_t *counts,
                             hipStream_t stream);

}  // namespace sd
