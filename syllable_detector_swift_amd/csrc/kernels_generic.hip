// kernels_generic.hip -- the always-applicable engine: any power-of-two FFT size, any
// window/gap/overlap, any input/output processing chain, any layer sizes.
//
// It materialises the sliced spectrogram [C][J][F] in HBM and evaluates the network per
// sliding window with no algebraic folding, i.e. it is the direct data-parallel restatement
// of   extractPower (CircularShortTimeFourierTransform.swift:280-337)
//   -> processFourierData (SyllableDetector.swift:134-151)
//   -> processNewValue / NeuralNet.apply (SyllableDetector.swift:153-217, NeuralNet.swift:294-326).
// The fused engine (kernels_fused.hip) is the fast path; this one is its fallback and the
// producer of syldet_spectrogram_device.
//
// gfx950 only: wave = 64 lanes, 256-thread workgroups.

#include "generic_eval.hpp"

namespace sd {

namespace {

using namespace generic_dev;

constexpr int kBlock = 256;

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// ------------------------------------------------------------------------------------
// 1024-point frames (BASELINE configs[2]): the 512-point complex FFT of the packed real frame as 8 x 8 x 8,
// one wave per frame, eight points per lane, three radix-8 passes in registers with two transposes through a
// wave-private 4.5 KB of LDS -- no workgroup barriers, 6 LDS round trips per point instead of 18.
//   n = 64 a + b, k = c + 8 d:     X[c + 8 d] = sum_b W64^(b d) . W512^(b c) . sum_a z[64 a + b] W8^(a c)
//   b = 8 a' + b', d = c' + 8 d':  (64-point part) = sum_b' W8^(b' d') . W64^(b' c') . sum_a' y[c][8 a' + b'] W8^(a' c')
// Window, packing, real split and |X| are those of stft_generic_kernel.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void dft8(float2 (&v)[8])
{
    // radix-2 decimation in frequency: X[2m] from a = v[j] + v[j+4], X[2m+1] from b = (v[j] - v[j+4]) W8^j, each a 4-point DFT
    const float h = 0.70710678118654752f;
    float2 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        a[j] = make_float2(v[j].x + v[j + 4].x, v[j].y + v[j + 4].y);
        b[j] = make_float2(v[j].x - v[j + 4].x, v[j].y - v[j + 4].y);
    }
    b[1] = make_float2(h * (b[1].x + b[1].y), h * (b[1].y - b[1].x));       // (1 - i)/sqrt2
    b[2] = make_float2(b[2].y, -b[2].x);                                    // -i
    b[3] = make_float2(h * (b[3].y - b[3].x), -h * (b[3].x + b[3].y));      // (-1 - i)/sqrt2
    auto dft4 = [](const float2 (&u)[4], float2 &y0, float2 &y1, float2 &y2, float2 &y3) {
        const float2 p0 = make_float2(u[0].x + u[2].x, u[0].y + u[2].y), p1 = make_float2(u[0].x - u[2].x, u[0].y - u[2].y);
        const float2 q0 = make_float2(u[1].x + u[3].x, u[1].y + u[3].y);
        const float2 q1 = make_float2(u[1].y - u[3].y, -(u[1].x - u[3].x));   // (u1 - u3) . (-i)
        y0 = make_float2(p0.x + q0.x, p0.y + q0.y);
        y1 = make_float2(p1.x + q1.x, p1.y + q1.y);
        y2 = make_float2(p0.x - q0.x, p0.y - q0.y);
        y3 = make_float2(p1.x - q1.x, p1.y - q1.y);
    };
    dft4(a, v[0], v[2], v[4], v[6]);
    dft4(b, v[1], v[3], v[5], v[7]);
}

constexpr int kR8Frames = 8;          // frames per wave
constexpr int kR8Lds = 8 * 72;        // float2 per wave: rows of 64 (+8) / 8 x 8 rows of 8 (+1) / 512 in natural order

__global__ void __launch_bounds__(kBlock)
stft_r8_kernel(StftDesc d, const float *__restrict__ samples, int64_t stride, int64_t J, float *__restrict__ columns)
{
    __shared__ float2 lds_all[(kBlock / kWave) * kR8Lds];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    float2 *buf = lds_all + wave * kR8Lds;
    const int c = blockIdx.y;
    const float *chan = samples + (int64_t)c * stride;
    float *cols = columns + (int64_t)c * J * d.F;
    const int hi3 = lane >> 3, lo3 = lane & 7;
    auto w1024 = [&](int idx) {                                   // e^{-2 pi i idx / 1024} from the half table
        idx &= 1023;
        const float2 w = d.sw[idx & 511];
        return idx & 512 ? make_float2(-w.x, -w.y) : w;
    };
    // per-lane constants: window for points 64 a + lane, twiddles of the two inter-pass multiplications
    float wre[8], wim[8];
    float2 tw1[8], tw2[8];
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const int n0 = 2 * (64 * a + lane);
        wre[a] = n0 < d.W ? d.window[n0] : 0.0f;
        wim[a] = n0 + 1 < d.W ? d.window[n0 + 1] : 0.0f;
        tw1[a] = w1024(2 * lane * a);                             // W512^(b c), b = lane, c = a
        tw2[a] = w1024(16 * lo3 * a);                             // W64^(b' c'), b' = lane & 7, c' = a
    }
    // split twiddles of this lane's bins f = lane + 64 it: the first two in registers (bands of up to 128 bins: BASELINE
    // configs[2] has 116), the rest fetched when a wider band needs them
    float2 swr[2];
#pragma unroll
    for (int it = 0; it < 2; it++) swr[it] = lane + kWave * it < d.F ? d.sw[d.f0 + lane + kWave * it] : make_float2(0.0f, 0.0f);
    const int64_t j0 = ((int64_t)blockIdx.x * (kBlock / kWave) + wave) * kR8Frames;
    if (j0 >= J) return;
    // raw samples of a frame: points 64 a + lane, as (even, odd) pairs; the next frame is fetched while this one is transformed
    auto fetch = [&](int64_t j, float2 (&raw)[8]) {
        const float *x = chan + j * d.hop + d.gap;
        if ((reinterpret_cast<uintptr_t>(x) & 7) == 0 && d.W == 1024) {   // wave-uniform: whole aligned frames take 8-byte loads
#pragma unroll
            for (int a = 0; a < 8; a++) raw[a] = reinterpret_cast<const float2 *>(x)[64 * a + lane];
        } else {
#pragma unroll
            for (int a = 0; a < 8; a++) {
                const int n0 = 2 * (64 * a + lane);
                raw[a] = make_float2(n0 < d.W ? x[n0] : 0.0f, n0 + 1 < d.W ? x[n0 + 1] : 0.0f);
            }
        }
    };
    float2 nxt[8];
    fetch(j0, nxt);
    for (int r = 0; r < kR8Frames; r++) {
        const int64_t j = j0 + r;
        if (j >= J) return;                                       // wave-uniform
        float2 v[8];
#pragma unroll
        for (int a = 0; a < 8; a++)                               // window multiply (vDSP_vmul :311); zero pad (:110) via zero window entries
            v[a] = make_float2(nxt[a].x * wre[a], nxt[a].y * wim[a]);
        if (r + 1 < kR8Frames && j + 1 < J) fetch(j + 1, nxt);
        dft8(v);                                                  // over a -> index c
#pragma unroll
        for (int cc = 0; cc < 8; cc++) buf[cc * 72 + lane] = cc ? cmul(v[cc], tw1[cc]) : v[cc];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = buf[hi3 * 72 + 8 * a + lo3];        // lane = (c, b'): y[c][8 a' + b']
        __builtin_amdgcn_wave_barrier();
        dft8(v);                                                  // over a' -> index c'
#pragma unroll
        for (int cc = 0; cc < 8; cc++) buf[(hi3 * 8 + cc) * 9 + lo3] = cc ? cmul(v[cc], tw2[cc]) : v[cc];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = buf[(hi3 * 8 + lo3) * 9 + a];       // lane = (c, c'): z[c][c'][b']
        __builtin_amdgcn_wave_barrier();
        dft8(v);                                                  // over b' -> index d'
#pragma unroll
        for (int dd = 0; dd < 8; dd++) buf[hi3 + 8 * lo3 + 64 * dd] = v[dd];   // Z[c + 8 c' + 64 d'], natural order
        __builtin_amdgcn_wave_barrier();
        // real split + magnitude for the band only; Nyquist is dropped (:323)
        auto split_bin = [&](int f, float2 w) {
            const int k = d.f0 + f;
            float re2, im2;
            if (k == 0) {
                const float2 z0 = buf[0];
                re2 = 2.0f * (z0.x + z0.y);
                im2 = 0.0f;
            } else {
                const float2 zk = buf[k], zm = buf[512 - k];
                const float ar = zk.x + zm.x, ai = zk.y - zm.y;
                const float br = zk.x - zm.x, bi = zk.y + zm.y;
                const float tr = br * w.x - bi * w.y, ti = br * w.y + bi * w.x;
                re2 = ar + ti;
                im2 = ai - tr;
            }
            const float p = re2 * re2 + im2 * im2;
            // zvmags/4 :270-274, zvabs/2 :329-333 (the hardware square root: 1 ulp, and NaN / inf / 0 as the library's)
            cols[j * d.F + f] = d.power_mode ? p * 0.25f : __builtin_amdgcn_sqrtf(p) * 0.5f;
        };
#pragma unroll
        for (int it = 0; it < 2; it++)
            if (lane + kWave * it < d.F) split_bin(lane + kWave * it, swr[it]);
        for (int f = lane + 2 * kWave; f < d.F; f += kWave) split_bin(f, d.sw[d.f0 + f]);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------
// STFT: one frame per group of TPF = min(256, M/4) threads, G = 256/TPF frames per pass.
// Packed real FFT: z[m] = xw[2m] + i xw[2m+1] (the vDSP_ctoz step :314-316), M-point
// complex Stockham autosort through two LDS buffers -- radix-4 stages, one radix-2 stage
// last when log2 M is odd (half the barriers and LDS round trips of a radix-2 chain) --
// then the real split restricted to the band:
//   2X[k] = (Z[k] + conj Z[M-k]) - i e^{-2 pi i k/N} (Z[k] - conj Z[M-k]).
// ------------------------------------------------------------------------------------
constexpr int kStftPasses = 8;   // frames per block = G * kStftPasses
constexpr int kStftTableM = 2048;  // up to this M the twiddle (M/2 float2) and window (<= 2M floats) tables live in LDS

__global__ void __launch_bounds__(kBlock)
stft_generic_kernel(StftDesc d, const float *__restrict__ samples, int64_t stride, int64_t J,
                    float *__restrict__ columns)
{
    extern __shared__ float2 lds[];
    const int M = d.M;
    const int half = M >> 1, quarter = M >> 2;
    const int TPF = quarter < kBlock ? (quarter < 1 ? 1 : quarter) : kBlock;
    const int G = kBlock / TPF;
    const int g = threadIdx.x / TPF;
    const int t = threadIdx.x - g * TPF;
    const int c = blockIdx.y;
    float2 *bufA = lds + (size_t)g * 2 * M;
    float2 *bufB = bufA + M;
    // small transforms keep the twiddle and window tables in LDS too (a global load per butterfly input costs a
    // memory latency in every stage); the launcher sizes the allocation to match
    const bool tables_in_lds = M <= kStftTableM;
    float2 *lds_tw = lds + (size_t)G * 2 * M;
    float2 *lds_sw = lds_tw + half;                             // split twiddles of the band: e^{-2 pi i (f0 + f) / N}
    float *lds_win = reinterpret_cast<float *>(lds_sw + d.F);
    if (tables_in_lds) {
        for (int i = threadIdx.x; i < half; i += kBlock) lds_tw[i] = d.tw[i];
        for (int i = threadIdx.x; i < d.F; i += kBlock) lds_sw[i] = d.sw[d.f0 + i];
        for (int i = threadIdx.x; i < d.W; i += kBlock) lds_win[i] = d.window[i];
        __syncthreads();
    }
    const float2 *twp = tables_in_lds ? lds_tw : d.tw;
    const float2 *swp = tables_in_lds ? lds_sw : d.sw + d.f0;
    const float *winp = tables_in_lds ? lds_win : d.window;
    const float *chan = samples + (int64_t)c * stride;
    float *cols = columns + (int64_t)c * J * d.F;

    // a frame whose threads all sit in one wave needs no workgroup barrier: a wave's LDS operations execute in order
    auto frame_sync = [&]() {
        if (TPF <= kWave) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
    };
    // every frame of this channel starts 8-byte aligned (block-uniform): sample pairs come in one load
    const bool pairs = (d.hop & 1) == 0 && (reinterpret_cast<uintptr_t>(chan + d.gap) & 7) == 0;
    for (int pass = 0; pass < kStftPasses; pass++) {
        const int64_t j = ((int64_t)blockIdx.x * kStftPasses + pass) * G + g;
        const bool valid = j < J;
        const float *x = chan + j * d.hop + d.gap;

        // window multiply (vDSP_vmul :311) + zero pad (:110) + even/odd packing: z[m] = xw[2m] + i xw[2m+1]
        auto packed = [&](int m) {
            const int n0 = 2 * m;
            float a = 0.0f, b = 0.0f;
            if (valid) {
                if (pairs && n0 + 1 < d.W) {                      // one 8-byte load per (even, odd) pair
                    const float2 s2 = reinterpret_cast<const float2 *>(x)[m];
                    a = s2.x * winp[n0];
                    b = s2.y * winp[n0 + 1];
                } else {
                    if (n0 < d.W) a = x[n0] * winp[n0];
                    if (n0 + 1 < d.W) b = x[n0 + 1] * winp[n0 + 1];
                }
            }
            return make_float2(a, b);
        };
        const bool first_from_memory = M >= 4;                    // the first radix-4 stage (all twiddles 1) reads the samples itself
        if (!first_from_memory) {
            for (int m = t; m < M; m += TPF) bufA[m] = packed(m);
            frame_sync();
        }

        float2 *in = bufA, *out = bufB;
        // twiddle e^{-2 pi i u / M} for u < M from the half table (tw[u], u < M/2)
        auto twiddle = [&](int u) {
            const float2 w = twp[u < half ? u : u - half];
            return u < half ? w : make_float2(-w.x, -w.y);
        };
        int Ns = 1;
        if (first_from_memory) {
            for (int jj = t; jj < quarter; jj += TPF) {
                const float2 v0 = packed(jj), v1 = packed(jj + quarter), v2 = packed(jj + 2 * quarter), v3 = packed(jj + 3 * quarter);
                const float2 p0 = make_float2(v0.x + v2.x, v0.y + v2.y), p1 = make_float2(v0.x - v2.x, v0.y - v2.y);
                const float2 q0 = make_float2(v1.x + v3.x, v1.y + v3.y);
                const float2 q1 = make_float2(v1.y - v3.y, -(v1.x - v3.x));   // (v1 - v3) . (-i)
                bufA[4 * jj] = make_float2(p0.x + q0.x, p0.y + q0.y);
                bufA[4 * jj + 1] = make_float2(p1.x + q1.x, p1.y + q1.y);
                bufA[4 * jj + 2] = make_float2(p0.x - q0.x, p0.y - q0.y);
                bufA[4 * jj + 3] = make_float2(p1.x - q1.x, p1.y - q1.y);
            }
            frame_sync();
            Ns = 4;
        }
        for (; Ns * 4 <= M; Ns <<= 2) {                           // radix-4 stages
            const int tw_stride = quarter / Ns;                  // e^{-2 pi i k / (4 Ns)} = twiddle(k * M/(4 Ns))
            for (int jj = t; jj < quarter; jj += TPF) {
                const int k = jj & (Ns - 1);
                const float2 v0 = in[jj];
                const float2 v1 = cmul(in[jj + quarter], twiddle(k * tw_stride));
                const float2 v2 = cmul(in[jj + 2 * quarter], twiddle(2 * k * tw_stride));
                const float2 v3 = cmul(in[jj + 3 * quarter], twiddle(3 * k * tw_stride));
                const float2 p0 = make_float2(v0.x + v2.x, v0.y + v2.y), p1 = make_float2(v0.x - v2.x, v0.y - v2.y);
                const float2 q0 = make_float2(v1.x + v3.x, v1.y + v3.y);
                const float2 q1 = make_float2(v1.y - v3.y, -(v1.x - v3.x));   // (v1 - v3) . (-i)
                const int j0 = ((jj - k) << 2) + k;
                out[j0] = make_float2(p0.x + q0.x, p0.y + q0.y);
                out[j0 + Ns] = make_float2(p1.x + q1.x, p1.y + q1.y);
                out[j0 + 2 * Ns] = make_float2(p0.x - q0.x, p0.y - q0.y);
                out[j0 + 3 * Ns] = make_float2(p1.x - q1.x, p1.y - q1.y);
            }
            frame_sync();
            float2 *tmp = in; in = out; out = tmp;
        }
        // log2 M odd: one radix-2 stage is left (Ns == M / 2).  It is not run: the real split below needs only the band's
        // Z[k] and Z[M - k], and computes those two butterflies itself from the stage's input
        const bool last_in_split = Ns < M;
        auto spectrum = [&](int u) {                              // Z[u]
            if (!last_in_split) return in[u];
            const int jj = u & (half - 1);
            const float2 v0 = in[jj];
            const float2 v1 = cmul(in[jj + half], twp[jj]);       // e^{-2 pi i jj / M}
            return u < half ? make_float2(v0.x + v1.x, v0.y + v1.y) : make_float2(v0.x - v1.x, v0.y - v1.y);
        };

        // real split + magnitude for the band only; Nyquist is dropped (:323)
        for (int f = t; f < d.F; f += TPF) {
            const int k = d.f0 + f;
            float re2, im2;
            if (k == 0) {
                const float2 z0 = spectrum(0);
                re2 = 2.0f * (z0.x + z0.y);
                im2 = 0.0f;
            } else {
                const float2 zk = spectrum(k), zm = spectrum(M - k);
                const float2 w = swp[f];
                const float ar = zk.x + zm.x, ai = zk.y - zm.y;
                const float br = zk.x - zm.x, bi = zk.y + zm.y;
                const float tr = br * w.x - bi * w.y, ti = br * w.y + bi * w.x;
                re2 = ar + ti;
                im2 = ai - tr;
            }
            const float p = re2 * re2 + im2 * im2;
            const float v = d.power_mode ? p * 0.25f : sqrtf(p) * 0.5f;   // zvmags/4 :270-274, zvabs/2 :329-333
            if (valid) cols[j * d.F + f] = v;
        }
        frame_sync();
    }
}

// ------------------------------------------------------------------------------------
// Network: one wave per evaluation, 4 evaluations per block pass.  Evaluation e reads
// the I = T*F contiguous floats columns[e*F .. e*F + I) (T columns, oldest first,
// SyllableDetector.swift:180-181), applies scaling (:184-212), the input functions in
// order (NeuralNet.swift:300-307), the layers (:310-313, :366-377), the output reverse
// maps (:316-323) and the threshold rule (SyllableDetector.swift:27-31 /
// TrackDetector.swift:72-77).
// ------------------------------------------------------------------------------------
constexpr int kMlpPasses = 8;    // evaluations per wave

__global__ void __launch_bounds__(kBlock)
mlp_generic_kernel(NetDesc n, int F, const float *__restrict__ columns, int64_t J, int64_t E,
                   float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    extern __shared__ float smem[];
    const int wave = threadIdx.x / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    const int c = blockIdx.y;
    float *bufA = smem + (size_t)wave * 2 * n.max_width;
    float *bufB = bufA + n.max_width;
    const float *cols = columns + (int64_t)c * J * F;

    for (int pass = 0; pass < kMlpPasses; pass++) {
        const int64_t e = ((int64_t)blockIdx.x * kMlpPasses + pass) * (kBlock / kWave) + wave;
        const bool valid = e < E;
        const float *src = cols + e * F;

        mlp_eval_wave(n, src, valid, bufA, bufB, lane, outputs ? outputs + (((int64_t)c * E) + e) * n.n_out : nullptr,
                      flags ? flags + (int64_t)c * E + e : nullptr);
    }
}

// ------------------------------------------------------------------------------------
// The same evaluation for networks whose first layer fits a wave's registers (inputs <= 64 KI, first layer <=
// HMAX wide, at most two layers, <= 8 outputs): one wave walks a run of consecutive evaluations with the input
// vector spread over its lanes (element i in lane i % 64, register i / 64) and the first-layer weights held in
// registers for the whole run -- no LDS, no barriers, reductions through DPP.  Operation order per evaluation is
// the reference's (scaling, input functions in file order, layers, reverse maps), nothing folded.
// ------------------------------------------------------------------------------------
constexpr int kSmallRun = 32;    // evaluations per wave

// WLDS: the first-layer rows live in LDS (one copy per workgroup, [row][element]) instead of registers -- for long input
// vectors, where 80 registers of weights would leave two waves per SIMD to hide every latency.
template <int KI, int HMAX, bool WLDS>
__global__ void __launch_bounds__(kBlock)
mlp_small_kernel(NetDesc n, int F, const float *__restrict__ columns, int64_t J, int64_t E,
                 float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int c = blockIdx.y;
    const float *cols = columns + (int64_t)c * J * F;
    const float *P = n.params;
    const int I = n.I;
    const DevLayer L0 = n.layers[0];
    const int H = L0.out;

    extern __shared__ float wlds[];                             // WLDS: [HMAX][KI * 64]
    float w[WLDS ? 1 : HMAX][WLDS ? 1 : KI];                    // else: first-layer rows, this lane's elements
    if (WLDS) {
        for (int i = threadIdx.x; i < HMAX * KI * kWave; i += kBlock) {
            const int h = i / (KI * kWave), e = i - h * (KI * kWave);
            wlds[i] = (h < H && e < I) ? P[L0.w + (size_t)h * I + e] : 0.0f;
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int h = 0; h < HMAX; h++)
#pragma unroll
            for (int k = 0; k < KI; k++) {
                const int i = lane + kWave * k;
                w[WLDS ? 0 : h][WLDS ? 0 : k] = (h < H && i < I) ? P[L0.w + (size_t)h * I + i] : 0.0f;
            }
    }
    const float b0 = lane < H ? P[L0.b + lane] : 0.0f;          // lane h finishes hidden unit h
    // the first affine input maps keep their per-element parameters in registers too
    constexpr int NCACHE = KI <= 5 ? 2 : 1;
    float axo[WLDS ? 1 : NCACHE][WLDS ? 1 : KI], aga[WLDS ? 1 : NCACHE][WLDS ? 1 : KI];
    float *alds = wlds + HMAX * KI * kWave;                     // WLDS: [offset | gain][KI * 64] of the first affine map
    if (WLDS) {
        int slot = 0;
        for (int q = 0; q < n.n_in_fns; q++) {
            if (n.in_fns[q].kind < 3) continue;
            if (slot++ == 0)
                for (int i = threadIdx.x; i < KI * kWave; i += kBlock) {
                    alds[i] = i < I ? P[n.in_fns[q].xoff + i] : 0.0f;
                    alds[KI * kWave + i] = i < I ? P[n.in_fns[q].gain + i] : 0.0f;
                }
        }
        __syncthreads();
    } else {
        int slot = 0;
        for (int q = 0; q < n.n_in_fns; q++) {
            if (n.in_fns[q].kind < 3 || slot >= NCACHE) continue;
#pragma unroll
            for (int u = 0; u < NCACHE; u++)
                if (u == slot)
#pragma unroll
                    for (int k = 0; k < KI; k++) {
                        const int i = lane + kWave * k;
                        axo[WLDS ? 0 : u][WLDS ? 0 : k] = i < I ? P[n.in_fns[q].xoff + i] : 0.0f;
                        aga[WLDS ? 0 : u][WLDS ? 0 : k] = i < I ? P[n.in_fns[q].gain + i] : 0.0f;
                    }
            slot++;
        }
    }

    const int64_t e0 = ((int64_t)blockIdx.x * (kBlock / kWave) + wave) * kSmallRun;
    if (e0 >= E) return;
    float xn[KI];                                               // the next evaluation's inputs, fetched one ahead
#pragma unroll
    for (int k = 0; k < KI; k++) xn[k] = lane + kWave * k < I ? cols[e0 * F + lane + kWave * k] : 0.0f;
    for (int r = 0; r < kSmallRun; r++) {
        const int64_t e = e0 + r;
        if (e >= E) return;                                     // wave-uniform
        float x[KI];
#pragma unroll
        for (int k = 0; k < KI; k++) x[k] = xn[k];
        if (r + 1 < kSmallRun && e + 1 < E) {
#pragma unroll
            for (int k = 0; k < KI; k++) xn[k] = lane + kWave * k < I ? cols[(e + 1) * F + lane + kWave * k] : 0.0f;
        }
        if (n.scaling != 0) {
#pragma unroll
            for (int k = 0; k < KI; k++) {
                const float v = n.scaling == 1 ? logf(x[k])                 // vvlogf, SyllableDetector.swift:207
                                               : 20.0f * log10f(x[k]);      // vDSP_vdbcon ref 1, amplitude flag :195
                x[k] = lane + kWave * k < I ? v : 0.0f;
            }
        }
        int wl = lane;
        if (WLDS) asm volatile("" : "+v"(wl));                  // keeps the LDS reads inside the loop (hoisted, they are 120 registers)
        int slot = 0;
        for (int q = 0; q < n.n_in_fns; q++) {
            const DevFn fn = n.in_fns[q];
            if (fn.kind == 0) {                                  // L2Normalize :47-59
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < KI; k++) s += x[k] * x[k];
                const float inv = 1.0f / sqrtf(wave_sum(s));
#pragma unroll
                for (int k = 0; k < KI; k++) x[k] = x[k] * inv;   // one division per vector (vDSP_vsdiv divides each; <= 1 ulp apart)
            } else if (fn.kind == 1) {                           // Normalize :69-96
                float mn = INFINITY, mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < KI; k++)
                    if (lane + kWave * k < I) { mn = fminf(mn, x[k]); mx = fmaxf(mx, x[k]); }
                mn = wave_min(mn);
                mx = wave_max(mx);
                const float range = mx - mn;
                const float slope = 2.0f / range, intercept = (0.0f - mn - mx) / range;
#pragma unroll
                for (int k = 0; k < KI; k++) x[k] = range == 0.0f ? -1.0f : x[k] * slope + intercept;
            } else if (fn.kind == 2) {                           // NormalizeStd :105-108 (population sigma)
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < KI; k++) s += x[k];
                const float mean = wave_sum(s) / (float)I;
                float qq = 0.0f;
#pragma unroll
                for (int k = 0; k < KI; k++)
                    if (lane + kWave * k < I) { const float dlt = x[k] - mean; qq += dlt * dlt; }
                const float isd = 1.0f / sqrtf(wave_sum(qq) / (float)I);
#pragma unroll
                for (int k = 0; k < KI; k++) x[k] = (x[k] - mean) * isd;
            } else {                                             // MapMinMax.apply :127-131, MapStd.apply :162-169
                bool done = false;
#pragma unroll
                for (int u = 0; u < NCACHE; u++)
                    if (u == slot) {
#pragma unroll
                        for (int k = 0; k < KI; k++)
                            x[k] = WLDS ? (x[k] - alds[k * kWave + wl]) * alds[(KI + k) * kWave + wl] + fn.y
                                        : (x[k] - axo[WLDS ? 0 : u][WLDS ? 0 : k]) * aga[WLDS ? 0 : u][WLDS ? 0 : k] + fn.y;
                        done = true;
                    }
                if (!done) {
#pragma unroll
                    for (int k = 0; k < KI; k++) {
                        const int i = lane + kWave * k;
                        if (i < I) x[k] = (x[k] - P[fn.xoff + i]) * P[fn.gain + i] + fn.y;
                    }
                }
                slot++;
            }
#pragma unroll
            for (int k = 0; k < KI; k++) x[k] = lane + kWave * k < I ? x[k] : 0.0f;   // padding stays out of every sum
        }
        // first layer: one dot product per row (every lane gets the sum); lane h then finishes unit h, so the
        // transfer function runs once per evaluation, and the activations come back as wave-uniform values
        float mine = 0.0f;
#pragma unroll
        for (int h = 0; h < HMAX; h++) {
            if (h < H) {                                        // wave-uniform
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < KI; k++) acc = fmaf(WLDS ? wlds[(h * KI + k) * kWave + wl] : w[WLDS ? 0 : h][WLDS ? 0 : k], x[k], acc);
                acc = wave_sum(acc);
                mine = lane == h ? acc : mine;
            }
        }
        const float act = transfer(L0.tf, mine + b0);
        float a[HMAX];
#pragma unroll
        for (int h = 0; h < HMAX; h++) a[h] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(act), h));
        // second layer (if any) and reverse maps: a handful of values, every lane computes them
        float y[8];
        int n_out = H;
#pragma unroll
        for (int o = 0; o < 8; o++) y[o] = o < HMAX ? a[o < HMAX ? o : 0] : 0.0f;
        if (n.n_layers == 2) {
            const DevLayer L1 = n.layers[1];
            n_out = L1.out;
#pragma unroll
            for (int o = 0; o < 8; o++) {
                float acc = 0.0f;
                if (o < n_out) {
#pragma unroll
                    for (int h = 0; h < HMAX; h++)
                        if (h < H) acc = fmaf(P[L1.w + o * H + h], a[h], acc);
                    acc = transfer(L1.tf, acc + P[L1.b + o]);
                }
                y[o] = acc;
            }
        }
        uint8_t hit = 0;
#pragma unroll
        for (int o = 0; o < 8; o++) {
            if (o < n_out) {
                float v = y[o];
                for (int q = 0; q < n.n_out_fns; q++) {          // reverse maps, NeuralNet.swift:137-142 / :175-180
                    const DevFn fn = n.out_fns[q];
                    v = (v - fn.y) / P[fn.gain + o] + P[fn.xoff + o];
                }
                if (lane == 0 && outputs) outputs[(((int64_t)c * E) + e) * n_out + o] = v;
                if (o == 0 || n.rule != 0) hit |= ((double)v >= n.thresholds[o]) ? 1 : 0;
            }
        }
        if (lane == 0 && flags) flags[(int64_t)c * E + e] = hit;
    }
}

// ------------------------------------------------------------------------------------
// mlp_small_kernel for the input chains the training script writes -- [l2normalize,] one affine map (mapminmax /
// mapstd) -- in front of a two-layer network with at most two outputs and at most one output map: the chain is a
// compile-time fact, every parameter of the second layer and of the output map sits in registers, and the run of
// evaluations is straight-line code (the interpretive kernel spends most of its time on scalar loads of the chain
// description, branches and per-evaluation parameter fetches).  Same operation order, nothing folded.
// ------------------------------------------------------------------------------------
template <int KI, int HMAX, bool WLDS, bool L2>
__global__ void __launch_bounds__(kBlock)
mlp_chain_kernel(NetDesc n, int F, const float *__restrict__ columns, int64_t J, int64_t E,
                 float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int c = blockIdx.y;
    const float *cols = columns + (int64_t)c * J * F;
    const float *P = n.params;
    const int I = n.I;
    const DevLayer L0 = n.layers[0], L1 = n.layers[1];
    const int H = L0.out, n_out = L1.out;
    const DevFn aff = n.in_fns[L2 ? 1 : 0];

    extern __shared__ float wlds[];                             // WLDS: [HMAX][KI * 64] rows, then offsets and gains [2][KI * 64]
    float *alds = wlds + HMAX * KI * kWave;
    float w[WLDS ? 1 : HMAX][WLDS ? 1 : KI], axo[WLDS ? 1 : KI], aga[WLDS ? 1 : KI];
    if (WLDS) {
        for (int i = threadIdx.x; i < HMAX * KI * kWave; i += kBlock) {
            const int h = i / (KI * kWave), e = i - h * (KI * kWave);
            wlds[i] = (h < H && e < I) ? P[L0.w + (size_t)h * I + e] : 0.0f;
        }
        for (int i = threadIdx.x; i < KI * kWave; i += kBlock) {
            alds[i] = i < I ? P[aff.xoff + i] : 0.0f;
            alds[KI * kWave + i] = i < I ? P[aff.gain + i] : 0.0f;
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int k = 0; k < KI; k++) {
            const int i = lane + kWave * k;
#pragma unroll
            for (int h = 0; h < HMAX; h++) w[WLDS ? 0 : h][WLDS ? 0 : k] = (h < H && i < I) ? P[L0.w + (size_t)h * I + i] : 0.0f;
            axo[WLDS ? 0 : k] = i < I ? P[aff.xoff + i] : 0.0f;
            aga[WLDS ? 0 : k] = i < I ? P[aff.gain + i] : 0.0f;
        }
    }
    const float b0 = lane < H ? P[L0.b + lane] : 0.0f;          // lane h finishes hidden unit h
    float w1[2][HMAX], b1[2], og[2], ox[2];
#pragma unroll
    for (int o = 0; o < 2; o++) {
#pragma unroll
        for (int h = 0; h < HMAX; h++) w1[o][h] = (o < n_out && h < H) ? P[L1.w + o * H + h] : 0.0f;
        b1[o] = o < n_out ? P[L1.b + o] : 0.0f;
        og[o] = (n.n_out_fns == 1 && o < n_out) ? P[n.out_fns[0].gain + o] : 1.0f;
        ox[o] = (n.n_out_fns == 1 && o < n_out) ? P[n.out_fns[0].xoff + o] : 0.0f;
    }
    const float oy = n.n_out_fns == 1 ? n.out_fns[0].y : 0.0f;
    const double thr0 = n.thresholds[0], thr1 = n_out > 1 ? n.thresholds[1] : 0.0;
    const int scaling = n.scaling, tf0 = L0.tf, tf1 = L1.tf;
    const bool any_rule = n.rule != 0;

    const int64_t e0 = ((int64_t)blockIdx.x * (kBlock / kWave) + wave) * kSmallRun;
    if (e0 >= E) return;
    float xn[KI];                                               // the next evaluation's inputs, fetched one ahead
#pragma unroll
    for (int k = 0; k < KI; k++) xn[k] = lane + kWave * k < I ? cols[e0 * F + lane + kWave * k] : 0.0f;
    for (int r = 0; r < kSmallRun; r++) {
        const int64_t e = e0 + r;
        if (e >= E) return;                                     // wave-uniform
        float x[KI];
#pragma unroll
        for (int k = 0; k < KI; k++) x[k] = xn[k];
        if (r + 1 < kSmallRun && e + 1 < E) {
#pragma unroll
            for (int k = 0; k < KI; k++) xn[k] = lane + kWave * k < I ? cols[(e + 1) * F + lane + kWave * k] : 0.0f;
        }
        if (scaling != 0) {
#pragma unroll
            for (int k = 0; k < KI; k++) {
                const float v = scaling == 1 ? logf(x[k])                   // vvlogf, SyllableDetector.swift:207
                                             : 20.0f * log10f(x[k]);        // vDSP_vdbcon ref 1, amplitude flag :195
                x[k] = lane + kWave * k < I ? v : 0.0f;
            }
        }
        int wl = lane;
        if (WLDS) asm volatile("" : "+v"(wl));                  // keeps the LDS reads inside the loop (hoisted, they are 120 registers)
        if (L2) {                                               // L2Normalize, NeuralNet.swift:47-59
            float ss = 0.0f;
#pragma unroll
            for (int k = 0; k < KI; k++) ss += x[k] * x[k];
            const float inv = 1.0f / sqrtf(wave_sum(ss));
#pragma unroll
            for (int k = 0; k < KI; k++) x[k] = x[k] * inv;     // one division per vector (vDSP_vsdiv divides each; <= 1 ulp apart)
        }
#pragma unroll
        for (int k = 0; k < KI; k++)                            // MapMinMax.apply :127-131, MapStd.apply :162-169
            x[k] = WLDS ? (x[k] - alds[k * kWave + wl]) * alds[(KI + k) * kWave + wl] + aff.y
                        : (x[k] - axo[WLDS ? 0 : k]) * aga[WLDS ? 0 : k] + aff.y;   // padding lanes meet zero weights below
        float mine = 0.0f;
#pragma unroll
        for (int h = 0; h < HMAX; h++) {
            if (h < H) {                                        // wave-uniform
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < KI; k++) acc = fmaf(WLDS ? wlds[(h * KI + k) * kWave + wl] : w[WLDS ? 0 : h][WLDS ? 0 : k], x[k], acc);
                acc = wave_sum(acc);
                mine = lane == h ? acc : mine;
            }
        }
        const float act = transfer(tf0, mine + b0);
        float y[2] = {0.0f, 0.0f};
#pragma unroll
        for (int h = 0; h < HMAX; h++) {
            const float a = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(act), h));
            y[0] = fmaf(w1[0][h], a, y[0]);                     // rows past H and outputs past n_out carry zero weights
            y[1] = fmaf(w1[1][h], a, y[1]);
        }
        uint8_t hit = 0;
#pragma unroll
        for (int o = 0; o < 2; o++) {
            if (o < n_out) {                                    // wave-uniform
                float v = transfer(tf1, y[o] + b1[o]);         // vDSP_mmul, then the bias (NeuralNet.swift:366-377)
                if (n.n_out_fns == 1) v = (v - oy) / og[o] + ox[o];   // reverse map, NeuralNet.swift:137-142 / :175-180
                if (lane == 0 && outputs) outputs[(((int64_t)c * E) + e) * n_out + o] = v;
                if (o == 0 || any_rule) hit |= ((double)v >= (o == 0 ? thr0 : thr1)) ? 1 : 0;
            }
        }
        if (lane == 0 && flags) flags[(int64_t)c * E + e] = hit;
    }
}

// ------------------------------------------------------------------------------------
// Detection sample numbers with debounce (TrackDetector.swift:39-43, :65-100): a greedy
// scan along time, one wave per channel, 64 flags per step; the wave skips ahead with
// ballots so quiet stretches cost one load per 64 evaluations.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kWave)
detections_kernel(const uint8_t *__restrict__ flags, int64_t E, int64_t first_index, int64_t hop,
                  int64_t debounce_frames, int64_t *__restrict__ indices, int64_t capacity,
                  int64_t *__restrict__ counts)
{
    const int c = blockIdx.x;
    const int lane = threadIdx.x;
    const uint8_t *fl = flags + (int64_t)c * E;
    int64_t *out = indices ? indices + (int64_t)c * capacity : nullptr;
    int64_t until = -1;      // debounceUntil :30
    int64_t n = 0;
    for (int64_t e0 = 0; e0 < E; e0 += kWave) {
        const int64_t e = e0 + lane;
        const bool set = e < E && fl[e] != 0;
        const int64_t idx = first_index + e * hop;               // curOutput :67-68
        unsigned long long mask = __ballot(set && until < idx);  // hasDetection && debounceUntil < curOutput :80
        while (mask) {
            const int l = __ffsll((long long)mask) - 1;
            const int64_t hit = first_index + (e0 + l) * hop;
            if (lane == 0 && out && n < capacity) out[n] = hit;
            n++;
            until = hit + debounce_frames;                       // :99
            const unsigned long long later = (l == 63) ? 0ull : (~0ull << (l + 1));
            mask = __ballot(set && until < idx) & later;
        }
    }
    if (lane == 0 && counts) counts[c] = n;
}

}  // namespace

hipError_t launch_stft_generic(const StftDesc &d, const float *samples, int64_t stride, int C, int64_t J,
                               float *columns, hipStream_t stream)
{
    if (J <= 0 || C <= 0) return hipSuccess;
    if (d.M == 512) {                              // 1024-point frames: the radix-8 wave-per-frame kernel
        const int64_t per_block = (int64_t)(kBlock / kWave) * kR8Frames;
        dim3 grid((unsigned)((J + per_block - 1) / per_block), (unsigned)C);
        hipLaunchKernelGGL(stft_r8_kernel, grid, dim3(kBlock), 0, stream, d, samples, stride, J, columns);
        return hipGetLastError();
    }
    const int quarter = d.M / 4;
    const int TPF = quarter < kBlock ? (quarter < 1 ? 1 : quarter) : kBlock;
    const int G = kBlock / TPF;
    const size_t lds = (size_t)G * 2 * d.M * sizeof(float2) + (d.M <= kStftTableM ? (size_t)(d.M / 2 + d.F) * sizeof(float2) + (size_t)d.W * sizeof(float) : 0);
    const int64_t per_block = (int64_t)G * kStftPasses;
    dim3 grid((unsigned)((J + per_block - 1) / per_block), (unsigned)C);
    if (lds > 64 * 1024) {
        hipError_t st = hipFuncSetAttribute((const void *)stft_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (st != hipSuccess) return st;
    }
    hipLaunchKernelGGL(stft_generic_kernel, grid, dim3(kBlock), lds, stream, d, samples, stride, J, columns);
    return hipGetLastError();
}

hipError_t launch_mlp_generic(const NetDesc &n, int F, const float *columns, int C, int64_t J, int64_t E,
                              float *outputs, uint8_t *flags, hipStream_t stream)
{
    if (E <= 0 || C <= 0) return hipSuccess;
    // small first layer: the register-resident kernel (not for chains without a normaliser: their first layer is summed in fp64,
    // which only the any-shape kernel does -- generic_eval.hpp)
    const bool normalised = n.n_in_fns > 0 && n.in_fns[0].kind < 3;
    if (normalised && n.n_layers >= 1 && n.n_layers <= 2 && n.n_out <= 8 && (n.n_layers == 1 || n.layers[1].in == n.layers[0].out)) {
        const int H = n.layers[0].out;
        const int64_t per_block = (int64_t)(kBlock / kWave) * kSmallRun;
        dim3 grid((unsigned)((E + per_block - 1) / per_block), (unsigned)C);
        // the chains the training script writes get the straight-line kernel: [l2normalize,] one affine map, two layers,
        // at most two outputs, at most one output map
        const bool affine_last = n.n_in_fns >= 1 && n.in_fns[n.n_in_fns - 1].kind >= 3;
        const bool chain_l2 = n.n_in_fns == 2 && n.in_fns[0].kind == 0 && affine_last;
        const bool chain_ok = n.n_layers == 2 && n.n_out <= 2 && n.n_out_fns <= 1 && (chain_l2 || (n.n_in_fns == 1 && affine_last));
        if (n.I <= 64 * 5 && H <= 8) {
            if (chain_ok && H <= 4 && chain_l2) hipLaunchKernelGGL((mlp_chain_kernel<5, 4, false, true>), grid, dim3(kBlock), 0, stream, n, F, columns, J, E, outputs, flags);
            else if (chain_ok && H <= 4) hipLaunchKernelGGL((mlp_chain_kernel<5, 4, false, false>), grid, dim3(kBlock), 0, stream, n, F, columns, J, E, outputs, flags);
            else if (chain_ok && chain_l2) hipLaunchKernelGGL((mlp_chain_kernel<5, 8, false, true>), grid, dim3(kBlock), 0, stream, n, F, columns, J, E, outputs, flags);
            else if (chain_ok) hipLaunchKernelGGL((mlp_chain_kernel<5, 8, false, false>), grid, dim3(kBlock), 0, stream, n, F, columns, J, E, outputs, flags);
            else hipLaunchKernelGGL((mlp_small_kernel<5, 8, false>), grid, dim3(kBlock), 0, stream, n, F, columns, J, E, outputs, flags);
            return hipGetLastError();
        }
        if (n.I <= 64 * 20 && H <= 4) {
            const size_t lds = (4 + 2) * 20 * kWave * sizeof(float);
            if (chain_ok && chain_l2) hipLaunchKernelGGL((mlp_chain_kernel<20, 4, true, true>), grid, dim3(kBlock), lds, stream, n, F, columns, J, E, outputs, flags);
            else if (chain_ok) hipLaunchKernelGGL((mlp_chain_kernel<20, 4, true, false>), grid, dim3(kBlock), lds, stream, n, F, columns, J, E, outputs, flags);
            else hipLaunchKernelGGL((mlp_small_kernel<20, 4, true>), grid, dim3(kBlock), lds, stream, n, F, columns, J, E, outputs, flags);
            return hipGetLastError();
        }
    }
    const size_t lds = (size_t)(kBlock / kWave) * 2 * n.max_width * sizeof(float);
    const int64_t per_block = (int64_t)(kBlock / kWave) * kMlpPasses;
    dim3 grid((unsigned)((E + per_block - 1) / per_block), (unsigned)C);
    if (lds > 64 * 1024) {
        hipError_t st = hipFuncSetAttribute((const void *)mlp_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (st != hipSuccess) return st;
    }
    hipLaunchKernelGGL(mlp_generic_kernel, grid, dim3(kBlock), lds, stream, n, F, columns, J, E, outputs, flags);
    return hipGetLastError();
}

hipError_t launch_detections(const uint8_t *flags, int C, int64_t E, int64_t first_index, int64_t hop,
                             int64_t debounce_frames, int64_t *indices, int64_t capacity, int64_t *counts,
                             hipStream_t stream)
{
    if (C <= 0) return hipSuccess;
    hipLaunchKernelGGL(detections_kernel, dim3((unsigned)C), dim3(kWave), 0, stream, flags, E, first_index, hop,
                       debounce_frames, indices, capacity, counts);
    return hipGetLastError();
}

}  // namespace sd
