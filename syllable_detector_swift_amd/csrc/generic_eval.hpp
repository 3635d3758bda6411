// generic_eval.hpp -- the unfolded per-evaluation network of the generic engine, shared by mlp_generic_kernel
// (kernels_generic.hip) and the exact recomputation of guarded evaluations (kernels_fixup.hip).
// gfx950 only.  wave = 64.
#pragma once

#include "kernels.hpp"

namespace sd {
namespace generic_dev {

constexpr int kWave = 64;

// Wave-wide reductions in registers: quad / half-row / row steps through DPP, rows and halves through gfx950's
// v_permlane16_swap / v_permlane32_swap (with both operands = x the two results are x and its partner).  Every
// lane ends up with the result.
template <typename Op>
__device__ __forceinline__ float wave_reduce(float v, Op op)
{
    v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xF, 0xF, false)));    // lane ^ 1
    v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4E, 0xF, 0xF, false)));    // lane ^ 2
    v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x141, 0xF, 0xF, false)));   // the other quad of the half row
    v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x140, 0xF, 0xF, false)));   // the other half of the row
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ double wave_sum_f64(double v)
{
    for (int m = 1; m < kWave; m <<= 1) v += __shfl_xor(v, m, kWave);
    return v;
}
__device__ __forceinline__ float wave_min(float v) { return wave_reduce(v, [](float a, float b) { return fminf(a, b); }); }
__device__ __forceinline__ float wave_max(float v) { return wave_reduce(v, [](float a, float b) { return fmaxf(a, b); }); }

__device__ __forceinline__ float transfer(int tf, float x)
{
    switch (tf) {
    case 0: return tanhf(x);                         // TanSig  NeuralNet.swift:189-194
    case 1: return 1.0f / (expf(-x) + 1.0f);         // LogSig  :196-215
    case 3: return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);   // SatLin  :223-228 (NaN stays NaN)
    default: return x;                               // PureLin :217-221
    }
}

// One evaluation by one wave, in the reference's operation order (nothing folded): scaling (SyllableDetector.swift:184-212),
// the input functions in file order (NeuralNet.swift:300-307), the layers (:310-313, :366-377), the reverse output maps
// (:316-323), the threshold rule (SyllableDetector.swift:27-31 / TrackDetector.swift:72-77).  `src`: the I = T * F column
// values of the window, oldest column first (global or LDS); bufA / bufB: this wave's two max_width-float buffers in LDS.
// Contains workgroup barriers: every wave of the workgroup calls it the same number of times (valid = false: a dry run).
__device__ __forceinline__ void mlp_eval_wave(const NetDesc &n, const float *src, bool valid, float *bufA, float *bufB, int lane,
                                              float *out, uint8_t *flag)
{
    const float *P = n.params;
    for (int i = lane; i < n.I; i += kWave) {
        float v = valid ? src[i] : 1.0f;
        if (n.scaling == 1) v = logf(v);                    // vvlogf, SyllableDetector.swift:207
        else if (n.scaling == 2) v = 20.0f * log10f(v);     // vDSP_vdbcon ref 1, amplitude flag :195
        bufA[i] = v;
    }
    __syncthreads();

    for (int k = 0; k < n.n_in_fns; k++) {
        const DevFn fn = n.in_fns[k];
        if (fn.kind == 0) {                                  // L2Normalize :47-59
            float s = 0.0f;
            for (int i = lane; i < n.I; i += kWave) s += bufA[i] * bufA[i];
            s = sqrtf(wave_sum(s));
            for (int i = lane; i < n.I; i += kWave) bufA[i] = bufA[i] / s;
        } else if (fn.kind == 1) {                           // Normalize :69-96
            float mn = INFINITY, mx = -INFINITY;
            for (int i = lane; i < n.I; i += kWave) { mn = fminf(mn, bufA[i]); mx = fmaxf(mx, bufA[i]); }
            mn = wave_min(mn);
            mx = wave_max(mx);
            const float range = mx - mn;
            if (range == 0.0f) {
                for (int i = lane; i < n.I; i += kWave) bufA[i] = -1.0f;
            } else {
                const float slope = 2.0f / range, intercept = (0.0f - mn - mx) / range;
                for (int i = lane; i < n.I; i += kWave) bufA[i] = bufA[i] * slope + intercept;
            }
        } else if (fn.kind == 2) {                           // NormalizeStd :105-108 (population sigma)
            float s = 0.0f;
            for (int i = lane; i < n.I; i += kWave) s += bufA[i];
            const float mean = wave_sum(s) / (float)n.I;
            float q = 0.0f;
            for (int i = lane; i < n.I; i += kWave) { const float dlt = bufA[i] - mean; q += dlt * dlt; }
            const float sd = sqrtf(wave_sum(q) / (float)n.I);
            for (int i = lane; i < n.I; i += kWave) bufA[i] = (bufA[i] - mean) / sd;
        } else if (fn.kind == 3) {                           // MapMinMax.apply :127-131
            for (int i = lane; i < n.I; i += kWave)
                bufA[i] = (bufA[i] - P[fn.xoff + i]) * P[fn.gain + i] + fn.y;
        } else {                                             // MapStd.apply :162-169
            for (int i = lane; i < n.I; i += kWave)
                bufA[i] = (bufA[i] - P[fn.xoff + i]) * P[fn.gain + i] + fn.y;
        }
        __syncthreads();
    }

    // No normaliser in front of the network: the first layer meets the columns at the recording's level, and a unit's sum can be
    // a small difference of large terms (columns of 3000 under weights of both signs: terms of +-180 that cancel to O(1)) -- in fp32
    // no order of summation holds 1e-5 there (round 5's sweep without level-dependent bars, draw 4125: 1.2e-5 here, 5e-6 for the
    // reference's own order).  Those sums are made in fp64: the products are exact, the sum is correctly rounded once.
    // Round 6: and so are the sums of every layer behind it for as long as nothing has bounded the values -- a PureLin layer hands the
    // level on (dB columns of -60 .. +20 through 7 -> 15 -> 2 linear layers: hidden values of hundreds that the second layer cancels to
    // 0.6; the 6000-draw sweep's draw 3104 under the one bar: 1.1e-5 in fp32 with EXACT columns).  TanSig, LogSig and SatLin bound
    // their outputs: the layers behind them are summed in fp32 as before.
    const bool first_f64 = n.n_in_fns == 0 || n.in_fns[0].kind >= 3;
    bool level_f64 = first_f64;
    float *cur = bufA, *nxt = bufB;
    for (int l = 0; l < n.n_layers; l++) {
        const DevLayer L = n.layers[l];
        const float *W = P + L.w;
        const bool this_f64 = level_f64;
        level_f64 = level_f64 && L.tf == 2;                  // (2: PureLin)
        if (this_f64) {
            for (int o = 0; o < L.out; o++) {
                double acc = 0.0;
                const float *wrow = W + (size_t)o * L.in;
                for (int i = lane; i < L.in; i += kWave) acc = fma((double)wrow[i], (double)cur[i], acc);
                acc = wave_sum_f64(acc);
                if (lane == 0) nxt[o] = transfer(L.tf, (float)(acc + (double)P[L.b + o]));
            }
        } else if (L.out < kWave) {
            // few outputs: the whole wave reduces one dot product at a time
            for (int o = 0; o < L.out; o++) {
                float acc = 0.0f;
                const float *wrow = W + (size_t)o * L.in;
                for (int i = lane; i < L.in; i += kWave) acc = fmaf(wrow[i], cur[i], acc);
                acc = wave_sum(acc);
                if (lane == 0) nxt[o] = transfer(L.tf, acc + P[L.b + o]);
            }
        } else {
            // many outputs: each lane owns whole rows
            for (int o = lane; o < L.out; o += kWave) {
                float acc = 0.0f;
                const float *wrow = W + (size_t)o * L.in;
                for (int i = 0; i < L.in; i++) acc = fmaf(wrow[i], cur[i], acc);
                nxt[o] = transfer(L.tf, acc + P[L.b + o]);
            }
        }
        __syncthreads();
        float *tmp = cur; cur = nxt; nxt = tmp;
    }

    // reverse maps, (y - yOff)/gain + xOffset, NeuralNet.swift:137-142 / :175-180
    for (int o = lane; o < n.n_out; o += kWave) {
        float v = cur[o];
        for (int k = 0; k < n.n_out_fns; k++) {
            const DevFn fn = n.out_fns[k];
            v = (v - fn.y) / P[fn.gain + o] + P[fn.xoff + o];
        }
        cur[o] = v;
        if (valid && out) out[o] = v;
    }
    __syncthreads();
    if (lane == 0 && valid && flag) {
        uint8_t hit = 0;
        const int lim = n.rule == 0 ? 1 : n.n_out;
        for (int o = 0; o < lim; o++) hit |= ((double)cur[o] >= n.thresholds[o]) ? 1 : 0;
        *flag = hit;
    }
    __syncthreads();
}

}  // namespace generic_dev
}  // namespace sd
