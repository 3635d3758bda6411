// kernels_wide.hip -- the wide-network engine (BASELINE configs[4]: a 4096-unit hidden layer): when the first
// layer is wide enough, many evaluations together make a dense GEMM, and that belongs on the matrix cores.
//
//   NeuralNet.apply (Common/NeuralNet.swift:294-326) for a batch of evaluations:
//     wide_prep_kernel   scaling + input functions per evaluation, exactly as the generic engine does them
//                        (fp32, the reference's operation order), result rounded to bf16:  Xn [evaluations][320]
//     wide_gemm_kernel   hidden = f0(W0 . x + b0) as  D[unit, evaluation] = W0[unit, :] . Xn[evaluation, :]  with
//                        v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulate); the second layer
//                        (a handful of outputs) is folded into the epilogue:  y[o] += w1[o, unit] * f0(.),
//                        then f1, reverse maps, threshold.  The hidden layer never exists in memory.
//
// Data flow of the GEMM: a wave owns 32 evaluations for the whole kernel and keeps their bf16 inputs in
// registers as MFMA B operands (20 k-steps x 4 registers); the weights stream through LDS in chunks of 32 hidden
// units (20 KB of A operands, packed in fragment order on the host; LDS-DMA, double-buffered), every chunk is read by all 16
// waves and used for 20 MFMAs each: 125 B/clk of LDS traffic per CU against 256 available; 4 waves per SIMD, so
// the matrix pipe has three other instruction streams to draw from while one wave is in its epilogue.  Roof: bf16 MFMA, 2.38 MFLOP per
// evaluation at H = 4096.
//
// Precision is bf16's (8-bit significands on inputs and weights): this engine is opt-in
// (SYLDET_ENGINE_WIDE_BF16), never chosen by AUTO, and its parity bar is stated separately (1e-2).
//
// gfx950 only.  wave = 64.

#include "kernels.hpp"

namespace sd {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kBlock = kWideBlock;            // 1024 threads = 16 waves
constexpr int kWave = 64;
constexpr int kKSteps = kWideK / 16;          // 20 k-steps of 16
constexpr int kChunkU4 = kWideChunkBytes / 16;     // 1320
constexpr int kChunkU4Pad = 21 * 64;               // an LDS buffer holds whole 64-element spans (the DMA writes base + 16 lane)
#ifndef SYLDET_WIDE_TW                              // (-DSYLDET_WIDE_TW=2: 19.15 ms against 18.27 on one box -- half the LDS bytes per
#define SYLDET_WIDE_TW 1                            // flop do not pay for having two waves a SIMD instead of four)
#endif
constexpr int kWideTilesPerWave = SYLDET_WIDE_TW;

__device__ __forceinline__ float wave_reduce_sum(float v)
{
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xF, 0xF, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4E, 0xF, 0xF, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x141, 0xF, 0xF, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x140, 0xF, 0xF, false));
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <typename Op>
__device__ __forceinline__ float wave_reduce(float v, Op op)
{
    v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xF, 0xF, false)));
    v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4E, 0xF, 0xF, false)));
    v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x141, 0xF, 0xF, false)));
    v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x140, 0xF, 0xF, false)));
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Transfer functions (NeuralNet.swift:185-228) through the hardware exp2 / rcp; NaN and the infinities fall out
// of the arithmetic (see kernels_fused.hip).
__device__ __forceinline__ float transfer_fast(int tf, float x)
{
    if (tf == 0) return fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.885390081777927f) + 1.0f), 1.0f);
    if (tf == 1) return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * -1.4426950408889634f) + 1.0f);
    if (tf == 3) return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
    return x;
}

// ------------------------------------------------------------------------------------
// Inputs of every evaluation, normalised, as bf16 rows of kWideK (zero padded).  One wave walks a run of
// evaluations with the vector spread over its lanes (element i in lane i % 64, register i / 64); the input
// chain is the generic engine's (SyllableDetector.swift:184-212, NeuralNet.swift:41-182).
// ------------------------------------------------------------------------------------
constexpr int kPrepRun = 32;
constexpr int kKI = kWideK / kWave;           // 5 registers per lane

__global__ void __launch_bounds__(256)
wide_prep_kernel(NetDesc n, int F, const float *__restrict__ columns, int64_t J, int64_t E, __bf16 *__restrict__ xn)
{
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int c = blockIdx.y;
    const float *cols = columns + (int64_t)c * J * F;
    const float *P = n.params;
    const int I = n.I;
    const int64_t e0 = ((int64_t)blockIdx.x * (256 / kWave) + wave) * kPrepRun;
    if (e0 >= E) return;
    float nxt[kKI];                                             // the next evaluation's inputs, fetched one ahead
#pragma unroll
    for (int k = 0; k < kKI; k++) nxt[k] = lane + kWave * k < I ? cols[e0 * F + lane + kWave * k] : 0.0f;
    for (int r = 0; r < kPrepRun; r++) {
        const int64_t e = e0 + r;
        if (e >= E) return;
        float x[kKI];
#pragma unroll
        for (int k = 0; k < kKI; k++) x[k] = nxt[k];
        if (r + 1 < kPrepRun && e + 1 < E) {
#pragma unroll
            for (int k = 0; k < kKI; k++) nxt[k] = lane + kWave * k < I ? cols[(e + 1) * F + lane + kWave * k] : 0.0f;
        }
        if (n.scaling != 0) {
#pragma unroll
            for (int k = 0; k < kKI; k++) {
                const float v = n.scaling == 1 ? logf(x[k]) : 20.0f * log10f(x[k]);
                x[k] = lane + kWave * k < I ? v : 0.0f;
            }
        }
        for (int q = 0; q < n.n_in_fns; q++) {
            const DevFn fn = n.in_fns[q];
            if (fn.kind == 0) {                                  // L2Normalize :47-59
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < kKI; k++) s += x[k] * x[k];
                s = sqrtf(wave_reduce_sum(s));
#pragma unroll
                for (int k = 0; k < kKI; k++) x[k] = x[k] / s;
            } else if (fn.kind == 1) {                           // Normalize :69-96
                float mn = INFINITY, mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < kKI; k++)
                    if (lane + kWave * k < I) { mn = fminf(mn, x[k]); mx = fmaxf(mx, x[k]); }
                mn = wave_reduce(mn, [](float a, float b) { return fminf(a, b); });
                mx = wave_reduce(mx, [](float a, float b) { return fmaxf(a, b); });
                const float range = mx - mn;
                const float slope = 2.0f / range, intercept = (0.0f - mn - mx) / range;
#pragma unroll
                for (int k = 0; k < kKI; k++) x[k] = range == 0.0f ? -1.0f : x[k] * slope + intercept;
            } else if (fn.kind == 2) {                           // NormalizeStd :105-108
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < kKI; k++) s += x[k];
                const float mean = wave_reduce_sum(s) / (float)I;
                float qq = 0.0f;
#pragma unroll
                for (int k = 0; k < kKI; k++)
                    if (lane + kWave * k < I) { const float dlt = x[k] - mean; qq += dlt * dlt; }
                const float sd = sqrtf(wave_reduce_sum(qq) / (float)I);
#pragma unroll
                for (int k = 0; k < kKI; k++) x[k] = (x[k] - mean) / sd;
            } else {                                             // MapMinMax.apply :127-131, MapStd.apply :162-169
#pragma unroll
                for (int k = 0; k < kKI; k++) {
                    const int i = lane + kWave * k;
                    if (i < I) x[k] = (x[k] - P[fn.xoff + i]) * P[fn.gain + i] + fn.y;
                }
            }
#pragma unroll
            for (int k = 0; k < kKI; k++) x[k] = lane + kWave * k < I ? x[k] : 0.0f;
        }
        __bf16 *dst = xn + ((int64_t)c * E + e) * kWideK;
#pragma unroll
        for (int k = 0; k < kKI; k++) dst[lane + kWave * k] = (__bf16)x[k];      // v_cvt_pk_bf16_f32: round to nearest even
    }
}

// The same for the input chains the training script writes -- [l2normalize,] one affine map: the chain is a
// compile-time fact and the map's parameters sit in registers (see mlp_chain_kernel in kernels_generic.hip).
template <bool L2>
__global__ void __launch_bounds__(256)
wide_prep_chain_kernel(NetDesc n, int F, const float *__restrict__ columns, int64_t J, int64_t E, __bf16 *__restrict__ xn)
{
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int c = blockIdx.y;
    const float *cols = columns + (int64_t)c * J * F;
    const float *P = n.params;
    const int I = n.I, scaling = n.scaling;
    const DevFn aff = n.in_fns[L2 ? 1 : 0];
    float axo[kKI], aga[kKI];
#pragma unroll
    for (int k = 0; k < kKI; k++) {
        const int i = lane + kWave * k;
        axo[k] = i < I ? P[aff.xoff + i] : 0.0f;
        aga[k] = i < I ? P[aff.gain + i] : 0.0f;
    }
    const int64_t e0 = ((int64_t)blockIdx.x * (256 / kWave) + wave) * kPrepRun;
    if (e0 >= E) return;
    float nxt[kKI];
#pragma unroll
    for (int k = 0; k < kKI; k++) nxt[k] = lane + kWave * k < I ? cols[e0 * F + lane + kWave * k] : 0.0f;
    for (int r = 0; r < kPrepRun; r++) {
        const int64_t e = e0 + r;
        if (e >= E) return;
        float x[kKI];
#pragma unroll
        for (int k = 0; k < kKI; k++) x[k] = nxt[k];
        if (r + 1 < kPrepRun && e + 1 < E) {
#pragma unroll
            for (int k = 0; k < kKI; k++) nxt[k] = lane + kWave * k < I ? cols[(e + 1) * F + lane + kWave * k] : 0.0f;
        }
        if (scaling != 0) {
#pragma unroll
            for (int k = 0; k < kKI; k++) {
                const float v = scaling == 1 ? logf(x[k]) : 20.0f * log10f(x[k]);
                x[k] = lane + kWave * k < I ? v : 0.0f;
            }
        }
        if (L2) {                                               // L2Normalize :47-59
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < kKI; k++) s += x[k] * x[k];
            const float inv = 1.0f / sqrtf(wave_reduce_sum(s));
#pragma unroll
            for (int k = 0; k < kKI; k++) x[k] = x[k] * inv;
        }
        __bf16 *dst = xn + ((int64_t)c * E + e) * kWideK;
#pragma unroll
        for (int k = 0; k < kKI; k++) {                         // MapMinMax.apply :127-131, MapStd.apply :162-169; zero padding
            const float v = lane + kWave * k < I ? (x[k] - axo[k]) * aga[k] + aff.y : 0.0f;
            dst[lane + kWave * k] = (__bf16)v;
        }
    }
}

// ------------------------------------------------------------------------------------
// The GEMM + epilogue.  MFMA operand layouts (v_mfma_f32_32x32x16_bf16):
//   A [32 units x 16 k]:   lane l holds unit l % 32, k = 8 (l / 32) + 0..7      (from LDS, host-packed)
//   B [16 k x 32 evals]:   lane l holds evaluation l % 32, k = 8 (l / 32) + 0..7 (registers, loaded once)
//   D [32 units x 32 evals]: lane l holds evaluation l % 32; register i holds unit 8 (i / 4) + 4 (l / 32) + i % 4
// ------------------------------------------------------------------------------------
// SIG: the hidden transfer function is TanSig or LogSig and the host has folded it into the tables -- weights and biases
// pre-scaled so that the accumulator (initialised with the bias) is the exponent, second-layer weights and bias rewritten
// so that  y += w1' / (2^acc + 1)  is all the epilogue does: exp2, add, rcp, fma per hidden value.
// TW: evaluation tiles per wave (1 is what ships).  With two, a wave owns 64 evaluations (160 registers of B operands), the
// workgroup has 8 waves instead of 16, and every A fragment fetched from LDS feeds two MFMAs: half the LDS bytes per flop.
// Measured slower (below).
template <int NOUT, bool SIG, int TW>
__global__ void __launch_bounds__(kBlock / TW, 1)
wide_gemm_kernel(WideDesc d, const uint4 *__restrict__ xn, int64_t NE, float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kWaves = kBlock / 64 / TW;
    uint4 *buf0 = reinterpret_cast<uint4 *>(smem), *buf1 = buf0 + kChunkU4Pad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    int64_t ev[TW];                                                           // this lane's evaluations (both halves)
#pragma unroll
    for (int t = 0; t < TW; t++) ev[t] = (int64_t)blockIdx.x * kWideTile + (wave * TW + t) * 32 + col;

    // this wave's evaluations as B operands, in registers for the whole kernel
    bf16x8 B[TW][kKSteps];
#pragma unroll
    for (int t = 0; t < TW; t++)
#pragma unroll
        for (int ks = 0; ks < kKSteps; ks++) {
            union { uint4 u; bf16x8 v; } b;
            b.u = ev[t] < NE ? xn[ev[t] * (kWideK / 8) + 2 * ks + half] : uint4{0, 0, 0, 0};
            B[t][ks] = b.v;
        }
    float ysum[TW][NOUT];
#pragma unroll
    for (int t = 0; t < TW; t++)
#pragma unroll
        for (int o = 0; o < NOUT; o++) ysum[t][o] = 0.0f;

    // Weight chunks go global -> LDS without passing through registers (LDS-DMA: lane l of a wave lands at base + 16 l);
    // chunk c+1 is in flight while chunk c is multiplied.
    auto fetch_chunk = [&](int ch, uint4 *dst) {
#pragma unroll
        for (int j = 0; j < 2 * TW; j++) {
            const int i0 = (wave + kWaves * j) * 64;              // this wave's 64 consecutive 16-byte elements
            if (i0 < kChunkU4) {
                const int i = i0 + lane < kChunkU4 ? i0 + lane : kChunkU4 - 1;   // the chunk ends inside the last span
                __builtin_amdgcn_global_load_lds(d.wpack + (size_t)ch * kChunkU4 + i, dst + i0, 16, 0, 0);
            }
        }
    };
    fetch_chunk(0, buf0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0): the LDS-DMA has landed
    __syncthreads();
    for (int ch = 0; ch < d.n_chunks; ch++) {
        const uint4 *cur = (ch & 1) ? buf1 : buf0;
        if (ch + 1 < d.n_chunks) fetch_chunk(ch + 1, (ch & 1) ? buf0 : buf1);
        const float *cst = reinterpret_cast<const float *>(cur + kKSteps * 64);   // this chunk's constants follow its fragments
        floatx16 acc[TW];
        {                                                         // the accumulators start at the bias: register i holds unit 8 (i/4) + 4 half + i%4
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float4 b0 = *reinterpret_cast<const float4 *>(cst + 8 * g + 4 * half);
#pragma unroll
                for (int t = 0; t < TW; t++) { acc[t][4 * g] = b0.x; acc[t][4 * g + 1] = b0.y; acc[t][4 * g + 2] = b0.z; acc[t][4 * g + 3] = b0.w; }
            }
        }
#pragma unroll
        for (int ks = 0; ks < kKSteps; ks++) {
            union { uint4 u; bf16x8 v; } a;
            a.u = cur[ks * 64 + lane];
#pragma unroll
            for (int t = 0; t < TW; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, B[t][ks], acc[t], 0, 0, 0);
        }
        // epilogue: transfer function, second-layer weights
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int u0 = 8 * g + 4 * half;                      // units u0 .. u0 + 3 live in registers 4g .. 4g + 3
            float4 w1[NOUT];
#pragma unroll
            for (int o = 0; o < NOUT; o++) w1[o] = o < d.n_out ? *reinterpret_cast<const float4 *>(cst + 32 + 32 * o + u0) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < TW; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float a0 = SIG ? __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc[t][4 * g + j]) + 1.0f)
                                         : transfer_fast(d.tf0, acc[t][4 * g + j]);
#pragma unroll
                    for (int o = 0; o < NOUT; o++) {
                        const float ww = j == 0 ? w1[o].x : (j == 1 ? w1[o].y : (j == 2 ? w1[o].z : w1[o].w));
                        ysum[t][o] = fmaf(a0, ww, ysum[t][o]);
                    }
                }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                       // the next chunk has landed (this wave's part)
        __syncthreads();
    }
    // the two lane halves hold disjoint units of the same evaluations
#pragma unroll
    for (int t = 0; t < TW; t++) {
#pragma unroll
        for (int o = 0; o < NOUT; o++) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(ysum[t][o]), __float_as_uint(ysum[t][o]), false, false);
            ysum[t][o] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        if (half == 0 && ev[t] < NE) {
            bool hit = false;
#pragma unroll
            for (int o = 0; o < NOUT; o++) {
                if (o >= d.n_out) break;
                float y = transfer_fast(d.tf1, ysum[t][o] + d.b1[o]);
                for (int q = 0; q < d.n_out_fns; q++) {           // reverse maps, NeuralNet.swift:137-142 / :175-180
                    const float *op = d.out_params + q * (1 + 2 * d.n_out);
                    y = (y - op[0]) / op[1 + o] + op[1 + d.n_out + o];
                }
                if (outputs) outputs[ev[t] * d.n_out + o] = y;
                if (o == 0 || d.rule == 1) hit = hit || ((double)y >= d.thresholds[o]);
            }
            if (flags) flags[ev[t]] = hit ? 1 : 0;
        }
    }
}


// ------------------------------------------------------------------------------------
// The same GEMM on v_mfma_f32_16x16x32_bf16 (round 3).  tools/energy_probe.py prices the 32x32x16 shape at 12 % more energy per
// flop and a 16 % lower clock under load than the 16x16x32 one (profiles/r03_energy_probe.txt), and this kernel runs on the
// power limit.  Same data flow: a wave owns 32 evaluations as B operands in registers (two column tiles of 16 x ten k-steps of
// 32: the same 80 registers), a chunk of 32 hidden units is two row tiles, every A fragment feeds two MFMAs.
//   A [16 units x 32 k]:   lane l holds unit l % 16 of its tile, k = 8 (l / 16) + 0..7   (from LDS, host-packed: [k-step][tile])
//   B [32 k x 16 evals]:   lane l holds evaluation l % 16 of its tile, k = 8 (l / 16) + 0..7
//   D [16 units x 16 evals]: lane l holds evaluation l % 16; register i holds unit 4 (l / 16) + i
// ------------------------------------------------------------------------------------
typedef float floatx4w __attribute__((ext_vector_type(4)));
#ifdef SYLDET_WIDE_X_TRACE      // (diagnostic build: evaluation tile 0's running output sum after every chunk, [workgroup][wave][chunk][lane])
__device__ float *g_wide_trace = nullptr;
#endif
// FRONT: the B operands are made here from the |X| columns (WideDesc::front) instead of being read from the prepared image.
// NWV: waves per workgroup.  16 (one workgroup of 512 evaluations a CU) or 8 (two workgroups of 256 a CU, each with its own
// chunk buffers and its own barrier: the two run out of phase, so that one's MFMA phase meets the other's epilogue --
// behind ONE barrier the 16 waves run every chunk in lockstep, 160 MFMAs with the vector unit waiting, then four epilogues
// with the matrix pipe idle; MEASUREMENTS R4.6).  The price: every workgroup streams the weights, so twice the L2 -> LDS bytes.
// STG (NWV = 8 only): the second half of the workgroup's waves -- the SIMD partners of the first half -- runs one epilogue behind: on
// every SIMD one wave is in a chunk's matrix instructions while its partner finishes the chunk before (MI355X_MICROARCH.md, "Two
// waves per SIMD", item 9).  A chunk then has readers during TWO barrier intervals, so the weights rotate through three buffers (the
// third laid over the front's stage, which is dead once the operands are in registers).  Every sum is made in the unstaggered order:
// the results are the unstaggered kernel's bit for bit.
// DMAB: the weight DMA through the compiler's builtin instead of the assembly statement (A/B and parity of the two forms:
// SYLDET_WIDE_DMA_BUILTIN=1; behind the builtin every later LDS read waits for the DMA, see fetch_chunk).
// TPW: evaluation tiles of 16 a wave.  2: 128 registers, four waves a SIMD (two workgroups of 8 a CU, or one of 16).  4 (with NWV = 8: ONE
// workgroup of 8 waves a CU, two waves a SIMD, 256 registers): a chunk's fragments, read from LDS once a wave, then feed twice the
// matrix instructions -- eight waves reading every chunk's 20 KB took as long as the chunk's matrix instructions.
// POLY (with SIG; SYLDET_WIDE_TANH_POLY=1, an A/B form): the hidden value is tanh_poly(acc) -- |acc| clamped to 3.3, then seven odd
// terms in packed fp32 (minimax with the clamp's tail: 1.36e-3 from tanh; tools/fit_tanh_poly.py) -- instead of 1 / (2^acc + 1)
// through exp2 and rcp: nine packed instructions and two clamps for two hidden values, no transcendental.  (In packed f16 the
// same polynomial is 1.5e-2 ... 3e-2 from tanh -- its high coefficients are f16 denormals and Horner's sums cancel -- so the
// 1e-2 bar rules that form out before any timing: MEASUREMENTS R6.3.)
template <int NOUT, bool SIG, bool FRONT, int NWV = 16, bool STG = false, bool DMAB = false, int TPW = 2, bool POLY = false>
__global__ void __launch_bounds__(64 * NWV, TPW == 4 ? 2 : (NWV == 16 ? 1 : 4))
wide_gemm16_kernel(WideDesc d, const uint4 *__restrict__ xn, const float *__restrict__ columns, int64_t J, int64_t E, int64_t NE,
                   float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kWaves = NWV, kK2 = kWideK / 32, kBl = 64 * NWV, kTl = 16 * TPW * NWV;
    uint4 *buf0 = reinterpret_cast<uint4 *>(smem), *buf1 = buf0 + kChunkU4Pad;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (a scalar: the DMA's LDS base and span tests stay out of the vector registers)
    const int n = lane & 15, g = lane >> 4;
    // FRONT: a workgroup's 512 evaluations are consecutive ones of ONE channel (grid y), so the columns under them are one
    // stretch of 511 F + I floats, staged through LDS once; otherwise evaluations are numbered through all channels
    const int64_t e_blk = (int64_t)blockIdx.x * kTl;        // (FRONT: within channel blockIdx.y)
    int64_t ev[TPW];
    bool ev_ok[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        const int64_t el = e_blk + wave * (16 * TPW) + 16 * t + n;
        ev[t] = FRONT ? (int64_t)blockIdx.y * E + el : el;
        ev_ok[t] = FRONT ? el < E : el < NE;
    }
    bf16x8 B[TPW][kK2];
    if (!FRONT) {
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int ks = 0; ks < kK2; ks++) {
                union { uint4 u; bf16x8 v; } b;
                b.u = ev_ok[t] ? xn[ev[t] * (kWideK / 8) + 4 * ks + g] : uint4{0, 0, 0, 0};
                B[t][ks] = b.v;
            }
    } else {
        // evaluation e of channel c: frames e .. e + T - 1 = I consecutive floats of the channel's [J][F] columns
        // (SyllableDetector.swift:158-181); this lane holds inputs 32 ks + 8 g + 0..7 of it.  The stretch under the workgroup's
        // evaluations goes through LDS (behind the two chunk buffers): coalesced loads once, then every lane picks its own
        float *stage = reinterpret_cast<float *>(smem + 2 * kChunkU4Pad * 16);
        const int F = d.F, I = d.I;
        const int span = (kTl - 1) * F + I;                 // (wide_front_stage_floats - 1)
        const float *chan = columns + (int64_t)blockIdx.y * J * F;
        const int64_t first = e_blk * F, limit = J * (int64_t)F;
        for (int i = tid; i < span; i += kBl) stage[i] = first + i < limit ? chan[first + i] : 0.0f;
        float *css = stage + span + 32;                           // [frames under the workgroup] a column's sum of squares (32 floats of slack: reads past I stay inside)
        __syncthreads();
        const int T = I / F;                                      // (I = F timeRange: SyllableDetector.swift:52-55)
        if (d.l2) {
            for (int f = tid; f < kTl + T - 1; f += kBl) {
                float a = 0.0f;
                for (int b = 0; b < F; b++) a = fmaf(stage[f * F + b], stage[f * F + b], a);
                css[f] = a;
            }
            __syncthreads();
        }
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const int el = wave * (16 * TPW) + 16 * t + n;
            int off = el * F + 8 * g;
            const int lim = I - 8 * g;
            float rinv = 1.0f;
            if (d.l2) {                                           // L2Normalize, NeuralNet.swift:47-59: the window's sum of squares from its columns'
                float ss = 0.0f;
                for (int tt = 0; tt < T; tt++) ss += css[el + tt];
                rinv = 1.0f / sqrtf(ss);                          // (silence: 0 * inf = NaN, as the reference's 0 / 0)
            }
            asm volatile("" : "+v"(off), "+v"(rinv));             // (the reads below stay below: hoisted, their 80 values would wait in scratch)
            const float *src = stage + off;
#pragma unroll
            for (int ks = 0; ks < kK2; ks++) {
                bf16x8 b;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float x = src[32 * ks + j];             // (always read; inputs past I are the next frames' values: zeroed)
                    b[j] = (__bf16)((32 * ks + j < lim ? x : 0.0f) * rinv);
                }
                union { bf16x8 v; uint4 u; } pk;                  // (finished here, four registers, before the next k-step's reads: left to
                pk.v = b;                                         // itself the compiler reads all 80 values first and parks them in scratch)
                asm volatile("" : "+v"(pk.u.x), "+v"(pk.u.y), "+v"(pk.u.z), "+v"(pk.u.w));
                B[t][ks] = pk.v;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                          // (nothing else uses the stage; the chunk buffers are next)
    }
    float ysum[TPW][NOUT];
    typedef float float2w __attribute__((ext_vector_type(2)));
    float2w ysum2[TPW][NOUT];                                     // (POLY: the sums of a lane's even and odd units apart, one packed multiply-add a pair)
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
        for (int o = 0; o < NOUT; o++) {
            ysum[t][o] = 0.0f;
            ysum2[t][o] = float2w{0.0f, 0.0f};
        }
    // (a buffer resource over the packed weights: a lane's address is one 32-bit offset, the chunk's a scalar)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t wp = (uint64_t)(uintptr_t)d.wpack;
    const u32x4 w_rs4 = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wp), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wp >> 32)) & 0xffffu,
                         (unsigned)(d.n_chunks * kChunkU4 * 16), 0x00020000u};
    const unsigned lane16 = (unsigned)lane * 16u;
    auto fetch_chunk = [&](int ch, uint4 *dst) {
#if defined(SYLDET_WIDE_X_DMATRAIL)                                   // (diagnostic: only the waves that pass a barrier last issue the DMA)
        constexpr int kIssuers = STG ? kWaves / 2 : kWaves;
        const int iw = STG ? wave - kWaves / 2 : wave;
        if (iw < 0) return;
#else
        constexpr int kIssuers = kWaves;
        const int iw = wave;
#endif
#pragma unroll
        for (int j = 0; j < (21 + kIssuers - 1) / kIssuers; j++) {
            const int i0 = (iw + kIssuers * j) * 64;
            if (i0 < kChunkU4 && DMAB) {
                const int i = i0 + lane < kChunkU4 ? i0 + lane : kChunkU4 - 1;   // (no range check on this path: the last span's lanes stay inside the chunk)
                __builtin_amdgcn_global_load_lds(d.wpack + (size_t)ch * kChunkU4 + i, dst + i0, 16, 0, 0);
            } else if (i0 < kChunkU4) {
                // (the chunk ends inside the last span: its lanes past the end read the next chunk's first bytes -- past the last chunk
                // the buffer's range check answers -- into the padding of the LDS buffer, which nobody reads; so every piece's lane
                // offset is the same register and everything else of its address is scalar)
                // (written out: behind the builtin the compiler takes the DMA for a store that may alias every later LDS read and
                // waits for it -- vmcnt(0) -- before the chunk's first ds_read, i.e. for the NEXT chunk's bytes at the top of
                // every chunk.  The wait that matters is the explicit one before the barrier.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"                       // ("clobber list contains reserved registers: m0" -- it does, on purpose)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"((unsigned)(uintptr_t)(dst + i0)), "v"(lane16), "s"(w_rs4), "s"((unsigned)ch * (unsigned)(kChunkU4 * 16) + (unsigned)i0 * 16u)
                             : "memory", "m0");
#pragma clang diagnostic pop
            }
        }
    };
    // a chunk: 20 MFMAs a tile pair from the chunk's fragments in LDS; then the hidden values and their share of the outputs
    auto multiply = [&](const uint4 *cur, floatx4w (&acc)[2][TPW]) {
        const float *cst = reinterpret_cast<const float *>(cur + kKSteps * 64);
#pragma unroll
        for (int ut = 0; ut < 2; ut++) {
            const float4 b0 = *reinterpret_cast<const float4 *>(cst + 16 * ut + 4 * g);
#pragma unroll
            for (int t = 0; t < TPW; t++) acc[ut][t] = floatx4w{b0.x, b0.y, b0.z, b0.w};
        }
#pragma unroll
        for (int ks = 0; ks < kK2; ks++) {
            union { uint4 u; bf16x8 v; } a0, a1;
            a0.u = cur[(2 * ks + 0) * 64 + lane];
            a1.u = cur[(2 * ks + 1) * 64 + lane];
#pragma unroll
            for (int t = 0; t < TPW; t++) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.v, B[t][ks], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.v, B[t][ks], acc[1][t], 0, 0, 0);
            }
        }
    };
#ifdef SYLDET_WIDE_X_TRACE
    // (diagnostic build: per chunk five records [lane][4] -- tile 0's hidden values of unit tiles 0 and 1, the second-layer weights the lane read for them, the running sums after the chunk -- [workgroup][wave][chunk][5][lane][4], stored through
    // inline assembly so that the compiler's wait counts stay what they are without it)
    const uint64_t tp = (uint64_t)(uintptr_t)g_wide_trace + ((((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * kWaves + wave) * (uint64_t)d.n_chunks * 5) * 1024;
    const u32x4 t_rs4 = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tp), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(tp >> 32)) & 0xffffu,
                         (unsigned)__builtin_amdgcn_readfirstlane(g_wide_trace ? d.n_chunks * 5 * 1024 : 0), 0x00020000u};
#endif
    auto finish = [&](const uint4 *cur, floatx4w (&acc)[2][TPW], int ch) {
        const float *cst = reinterpret_cast<const float *>(cur + kKSteps * 64);
#ifdef SYLDET_WIDE_X_PAD                                              // (diagnostic: idle states between the last matrix instruction and the first read of its result)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
        __builtin_amdgcn_sched_barrier(0);
#endif
        if constexpr (POLY) {
            static_assert(!POLY || SIG, "the polynomial stands for TanSig / LogSig");
            // tanh_poly on pairs (registers 0,1 and 2,3 of a result tile are neighbours): clamp, u = x x, Horner in u, times x
            constexpr float kC = 3.3f;
            const float2w c0 = {0.9934016466140747f, 0.9934016466140747f}, c1 = {-0.30040496587753296f, -0.30040496587753296f},
                          c2 = {0.08361941576004028f, 0.08361941576004028f}, c3 = {-0.015534450300037861f, -0.015534450300037861f},
                          c4 = {0.0017269821837544441f, 0.0017269821837544441f}, c5 = {-0.00010273736552335322f, -0.00010273736552335322f},
                          c6 = {2.5018941869348055e-06f, 2.5018941869348055e-06f};
#pragma unroll
            for (int ut = 0; ut < 2; ut++) {
                float2w hv[TPW][2];
#pragma unroll
                for (int t = 0; t < TPW; t++)
#pragma unroll
                    for (int pr = 0; pr < 2; pr++) {
                        const float2w x = {__builtin_amdgcn_fmed3f(acc[ut][t][2 * pr], -kC, kC), __builtin_amdgcn_fmed3f(acc[ut][t][2 * pr + 1], -kC, kC)};
                        const float2w u = x * x;
                        float2w q = __builtin_elementwise_fma(c6, u, c5);
                        q = __builtin_elementwise_fma(q, u, c4);
                        q = __builtin_elementwise_fma(q, u, c3);
                        q = __builtin_elementwise_fma(q, u, c2);
                        q = __builtin_elementwise_fma(q, u, c1);
                        q = __builtin_elementwise_fma(q, u, c0);
                        hv[t][pr] = q * x;
                    }
#pragma unroll
                for (int o = 0; o < NOUT; o++) {
                    const float4 w1 = *reinterpret_cast<const float4 *>(cst + 32 + 32 * o + 16 * ut + 4 * g);
                    const float2w wa = {w1.x, w1.y}, wb = {w1.z, w1.w};
#pragma unroll
                    for (int t = 0; t < TPW; t++) {
                        ysum2[t][o] = __builtin_elementwise_fma(wa, hv[t][0], ysum2[t][o]);      // (the weights first: see below)
                        ysum2[t][o] = __builtin_elementwise_fma(wb, hv[t][1], ysum2[t][o]);
                    }
                }
            }
            if (ch == d.n_chunks - 1) {
#pragma unroll
                for (int t = 0; t < TPW; t++)
#pragma unroll
                    for (int o = 0; o < NOUT; o++) {
                        // (the two halves of a pair as ONE plain addition, written out: left to the compiler, two tiles' horizontal sums become
                        // a v_pk_add_f32 that selects src1's high half into the low half -- the form tools/check_pk_opsel.py forbids)
                        float r;
                        asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(ysum2[t][o].x), "v"(ysum2[t][o].y));
                        ysum[t][o] = r;
                    }
            }
            return;
        }
#pragma unroll
        for (int ut = 0; ut < 2; ut++) {
#pragma unroll
            for (int t = 0; t < TPW; t++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[ut][t][j] = SIG ? __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc[ut][t][j]) + 1.0f) : transfer_fast(d.tf0, acc[ut][t][j]);
#ifdef SYLDET_WIDE_X_TRACE
            asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(acc[ut][0]), "v"(lane16), "s"(t_rs4), "s"((unsigned)(ch * 5 + ut) * 1024u) : "memory");
#endif
#pragma unroll
            for (int o = 0; o < NOUT; o++) {                      // (rows of unused outputs are zero in the table)
                const float4 w1 = *reinterpret_cast<const float4 *>(cst + 32 + 32 * o + 16 * ut + 4 * g);
#ifdef SYLDET_WIDE_X_TRACE
                if (o == 0) {
                    const floatx4w w1v = {w1.x, w1.y, w1.z, w1.w};
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(w1v), "v"(lane16), "s"(t_rs4), "s"((unsigned)(ch * 5 + 2 + ut) * 1024u) : "memory");
                }
#endif
#pragma unroll
                for (int t = 0; t < TPW; t++) {
                    // (the weight FIRST: the compiler packs the two tiles' multiply-adds into v_pk_fma_f32 and takes w1.y / w1.w -- the high
                    // registers of their pairs -- by operand selection; on src1 that selection loses the low half's product in lanes 48-63
                    // whenever another wave of the SIMD is in its matrix instructions (tools/ubench/pkfma_opsel.hip, MEASUREMENTS R5.1), on
                    // src0 it does not.  The build checks the ISA of every kernel for the src1 form: tools/check_pk_opsel.py.)
                    ysum[t][o] = fmaf(w1.x, acc[ut][t][0], ysum[t][o]);
                    ysum[t][o] = fmaf(w1.y, acc[ut][t][1], ysum[t][o]);
                    ysum[t][o] = fmaf(w1.z, acc[ut][t][2], ysum[t][o]);
                    ysum[t][o] = fmaf(w1.w, acc[ut][t][3], ysum[t][o]);
                }
            }
        }
#ifdef SYLDET_WIDE_X_TRACE
        {
            const floatx4w yv = {ysum[0][0], ysum[1][0], 0.0f, 0.0f};
            asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(yv), "v"(lane16), "s"(t_rs4), "s"((unsigned)(ch * 5 + 4) * 1024u) : "memory");
        }
#endif
    };
    if constexpr (!STG) {
        fetch_chunk(0, buf0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        for (int ch = 0; ch < d.n_chunks; ch++) {
            const uint4 *cur = (ch & 1) ? buf1 : buf0;
            if (ch + 1 < d.n_chunks) fetch_chunk(ch + 1, (ch & 1) ? buf0 : buf1);
            floatx4w acc[2][TPW];                                   // [unit tile][evaluation tile]: register i = unit 16 ut + 4 g + i
            multiply(cur, acc);
            finish(cur, acc, ch);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
        }
    } else {
        static_assert(!STG || (NWV == 8 && FRONT), "the staggered form is the two-workgroups-a-CU kernel's");
        // Chunk c lives in buffer c mod 3.  In barrier interval c (between barrier c and barrier c + 1) the leading waves read
        // chunk c, the trailing waves chunks c - 1 (its second-layer weights) and c, and everybody's share of chunk c + 1 is on its
        // way into buffer (c + 1) mod 3 = (c - 2) mod 3, whose last readers -- the trailing waves, in interval c - 1 -- retired
        // their reads (lgkmcnt(0)) before barrier c.  A wave waits for its own pieces of chunk c + 1 (vmcnt(0)) before barrier
        // c + 1; its readers are behind that barrier.
        auto bufp = [&](int b) { return buf0 + b * kChunkU4Pad; };    // (the third one is the front's stage)
        auto next = [](int b) { return b == 2 ? 0 : b + 1; };
        auto seal = [&]() {
            // (The compiler is free to move a chunk's register arithmetic across its barrier and to interleave the trailing waves'
            // epilogue with their next matrix instructions -- 15.78 ms against 15.96 with everything pinned to its interval
            // (-DSYLDET_WIDE_X_STRICT), 15.98 unstaggered, one box, profiles/r05_wide_stagger_ab.txt.  What it must not do is read LDS
            // across the barrier, and it cannot: the barrier is a fence for it.)
#ifdef SYLDET_WIDE_X_STRICT
            __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef SYLDET_WIDE_X_NOLGKM
            __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0) only
#else
            __builtin_amdgcn_s_waitcnt(0x0070);                   // vmcnt(0): my pieces of the next chunk have landed; lgkmcnt(0): my reads of this interval are done
#endif
            __syncthreads();
#ifdef SYLDET_WIDE_X_BAR2
            __builtin_amdgcn_s_waitcnt(0x0070);
            __syncthreads();
#endif
#ifdef SYLDET_WIDE_X_STRICT
            __builtin_amdgcn_sched_barrier(0);
#endif
        };
#if defined(SYLDET_WIDE_X_SETPRIO_TRAIL)                 // (round 6 experiments: a static priority for the trailing / the leading half)
        if (wave >= kWaves / 2) __builtin_amdgcn_s_setprio(1);
#elif defined(SYLDET_WIDE_X_SETPRIO_LEAD)
        if (wave < kWaves / 2) __builtin_amdgcn_s_setprio(1);
#endif
        const int nch = d.n_chunks;
        fetch_chunk(0, bufp(0));
        seal();
        if (wave < kWaves / 2) {
            int bi = 0;
            for (int ch = 0; ch < nch; ch++) {
                const int bn = next(bi);
                if (ch + 1 < nch) fetch_chunk(ch + 1, bufp(bn));
                floatx4w acc[2][TPW];
                multiply(bufp(bi), acc);
                finish(bufp(bi), acc, ch);
                    seal();
                bi = bn;
            }
        } else {
            floatx4w acc[2][TPW];                                   // (chunk c - 1's sums cross barrier c in registers)
            if (1 < nch) fetch_chunk(1, bufp(1));
            multiply(bufp(0), acc);
            seal();
            int bp = 0, bi = 1;
            for (int ch = 1; ch < nch; ch++) {
                const int bn = next(bi);
                if (ch + 1 < nch) fetch_chunk(ch + 1, bufp(bn));
                finish(bufp(bp), acc, ch - 1);
#ifdef SYLDET_WIDE_X_STRICT                                           // (diagnostic: the chunk before finished before this one's matrix instructions start)
                __builtin_amdgcn_sched_barrier(0);
#endif
                multiply(bufp(bi), acc);
                seal();
                bp = bi; bi = bn;
            }
            finish(bufp(bp), acc, nch - 1);
        }
    }
    // the four lane groups hold disjoint units of the same evaluations
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));                                // (recomputed: held across the loop they cost two spilled pairs)
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        const int64_t el = (int64_t)blockIdx.x * kTl + (tid2 >> 6) * (16 * TPW) + 16 * t + (tid2 & 15);
        ev[t] = FRONT ? (int64_t)blockIdx.y * E + el : el;
        ev_ok[t] = FRONT ? el < E : el < NE;
    }
#pragma unroll
    for (int t = 0; t < TPW; t++) {
#pragma unroll
        for (int o = 0; o < NOUT; o++) {
            auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(ysum[t][o]), __float_as_uint(ysum[t][o]), false, false);
            float v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
            ysum[t][o] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        if (g == 0 && ev_ok[t]) {
            bool hit = false;
#pragma unroll
            for (int o = 0; o < NOUT; o++) {
                if (o >= d.n_out) break;
                float y = transfer_fast(d.tf1, ysum[t][o] + d.b1[o]);
                for (int q = 0; q < d.n_out_fns; q++) {
                    const float *op = d.out_params + q * (1 + 2 * d.n_out);
                    y = (y - op[0]) / op[1 + o] + op[1 + d.n_out + o];
                }
                if (outputs) outputs[ev[t] * d.n_out + o] = y;
                if (o == 0 || d.rule == 1) hit = hit || ((double)y >= d.thresholds[o]);
            }
            if (flags) flags[ev[t]] = hit ? 1 : 0;
        }
    }
}


// ------------------------------------------------------------------------------------
// Round 6 (SYLDET_WIDE_M32=1, an A/B form): the staggered two-workgroups-a-CU GEMM on v_mfma_f32_32x32x16_bf16.  bench.py's counters
// say the 16x16x32 kernel is bound by the SIMD's ISSUE port (a chunk costs a wave ~1000 issue clocks -- 40 matrix instructions at 8,
// sixteen hidden values at ~20, reads, DMA pieces -- four waves a SIMD: 4000 against the matrix pipe's 2560), and a 32x32x16
// instruction holds the port for the same 8 clocks while it does twice the work: 19 instructions a chunk and wave instead of 40
// (19, not 20: 290 inputs fill 19 k-steps of 16 -- the 16x16x32 shape pads to 320), at the price of 12 % more energy per flop
// (profiles/r03_energy_probe.txt) under a kernel that already runs at 1.9-2.0 GHz.  A wave owns 32 evaluations as ONE column
// tile (19 or 20 x 4 registers of B operands), a chunk's 32 hidden units are one row tile:
//   A [32 units x 16 k]:   lane l holds unit l % 32, k = 8 (l / 32) + 0..7      (from LDS; the host's 32x32x16 packing)
//   B [16 k x 32 evals]:   lane l holds evaluation l % 32, k = 8 (l / 32) + 0..7
//   D [32 units x 32 evals]: lane l holds evaluation l % 32; register i holds unit 8 (i / 4) + 4 (l / 32) + i % 4
// One output, the front end (columns read here), 8 waves, staggered halves, three chunk buffers, the DMA as the assembly statement:
// the shipped kernel's structure (see wide_gemm16_kernel), nothing else instantiated.  KSN: k-steps that hold inputs.
template <bool SIG, int KSN>
__global__ void __launch_bounds__(512, 4)
wide_gemm32s_kernel(WideDesc d, const float *__restrict__ columns, int64_t J, int64_t E, int64_t NE, float *__restrict__ outputs,
                    uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kWaves = 8, kBl = 512, kTl = 256;
    uint4 *buf0 = reinterpret_cast<uint4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, g = lane >> 5;
    const int64_t e_blk = (int64_t)blockIdx.x * kTl;              // (within channel blockIdx.y)
    bf16x8 B[KSN];
    {
        // evaluation e of channel c: frames e .. e + T - 1 = I consecutive floats of the channel's [J][F] columns; this lane holds
        // inputs 16 ks + 8 g + 0..7 of evaluation wave 32 + n.  The stretch under the workgroup's 256 evaluations goes through LDS
        float *stage = reinterpret_cast<float *>(smem + 2 * kChunkU4Pad * 16);
        const int F = d.F, I = d.I;
        const int span = (kTl - 1) * F + I;
        const float *chan = columns + (int64_t)blockIdx.y * J * F;
        const int64_t first = e_blk * F, limit = J * (int64_t)F;
        for (int i = tid; i < span; i += kBl) stage[i] = first + i < limit ? chan[first + i] : 0.0f;
        float *css = stage + span + 32;
        __syncthreads();
        const int T = I / F;
        if (d.l2) {
            for (int f = tid; f < kTl + T - 1; f += kBl) {
                float a = 0.0f;
                for (int b = 0; b < F; b++) a = fmaf(stage[f * F + b], stage[f * F + b], a);
                css[f] = a;
            }
            __syncthreads();
        }
        const int el = wave * 32 + n;
        int off = el * F + 8 * g;
        const int lim = I - 8 * g;
        float rinv = 1.0f;
        if (d.l2) {
            float ss = 0.0f;
            for (int tt = 0; tt < T; tt++) ss += css[el + tt];
            rinv = 1.0f / sqrtf(ss);                              // (silence: 0 * inf = NaN, as the reference's 0 / 0)
        }
        asm volatile("" : "+v"(off), "+v"(rinv));
        const float *src = stage + off;
#pragma unroll
        for (int ks = 0; ks < KSN; ks++) {
            bf16x8 b;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float x = src[16 * ks + j];                 // (always read; inputs past I are the next frames' values: zeroed)
                b[j] = (__bf16)((16 * ks + j < lim ? x : 0.0f) * rinv);
            }
            union { bf16x8 v; uint4 u; } pk;
            pk.v = b;
            asm volatile("" : "+v"(pk.u.x), "+v"(pk.u.y), "+v"(pk.u.z), "+v"(pk.u.w));
            B[ks] = pk.v;
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    float ysum0 = 0.0f, ysum1 = 0.0f;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t wp = (uint64_t)(uintptr_t)d.wpack;
    const u32x4 w_rs4 = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wp), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wp >> 32)) & 0xffffu,
                         (unsigned)(d.n_chunks * kChunkU4 * 16), 0x00020000u};
    const unsigned lane16 = (unsigned)lane * 16u;
    auto fetch_chunk = [&](int ch, uint4 *dst) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int piece = wave + kWaves * j;                   // piece p = fragment of k-step p (p < 20), then the constants
            const int i0 = piece * 64;
            // (a k-step that holds no inputs is all zeros and is not multiplied: its piece stays where it is)
            if (i0 < kChunkU4 && !(piece >= KSN && piece < kKSteps)) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"((unsigned)(uintptr_t)(dst + i0)), "v"(lane16), "s"(w_rs4), "s"((unsigned)ch * (unsigned)(kChunkU4 * 16) + (unsigned)i0 * 16u)
                             : "memory", "m0");
#pragma clang diagnostic pop
            }
        }
    };
    auto multiply = [&](const uint4 *cur, floatx16 &acc) {
        const float *cst = reinterpret_cast<const float *>(cur + kKSteps * 64);
#pragma unroll
        for (int q = 0; q < 4; q++) {                              // the accumulators start at the bias: register 4 q + i holds unit 8 q + 4 g + i
            const float4 b0 = *reinterpret_cast<const float4 *>(cst + 8 * q + 4 * g);
            acc[4 * q] = b0.x; acc[4 * q + 1] = b0.y; acc[4 * q + 2] = b0.z; acc[4 * q + 3] = b0.w;
        }
#pragma unroll
        for (int ks = 0; ks < KSN; ks++) {
            union { uint4 u; bf16x8 v; } a;
            a.u = cur[ks * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, B[ks], acc, 0, 0, 0);
        }
    };
    auto finish = [&](const uint4 *cur, floatx16 &acc) {
        const float *cst = reinterpret_cast<const float *>(cur + kKSteps * 64);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 w1 = *reinterpret_cast<const float4 *>(cst + 32 + 8 * q + 4 * g);
            float h[4];
#pragma unroll
            for (int i = 0; i < 4; i++) h[i] = SIG ? __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc[4 * q + i]) + 1.0f) : transfer_fast(d.tf0, acc[4 * q + i]);
            // (the weight first: see wide_gemm16_kernel)
            float &y = (q & 1) ? ysum1 : ysum0;
            y = fmaf(w1.x, h[0], y);
            y = fmaf(w1.y, h[1], y);
            y = fmaf(w1.z, h[2], y);
            y = fmaf(w1.w, h[3], y);
        }
    };
    {
        // chunk c lives in buffer c mod 3 (the third over the front's stage); waves 4-7 run one epilogue behind waves 0-3: wide_gemm16_kernel, STG
        auto bufp = [&](int b) { return buf0 + b * kChunkU4Pad; };
        auto next = [](int b) { return b == 2 ? 0 : b + 1; };
        auto seal = [&]() {
            __builtin_amdgcn_s_waitcnt(0x0070);                   // vmcnt(0): my pieces of the next chunk have landed; lgkmcnt(0): my reads of this interval are done
            __syncthreads();
        };
        const int nch = d.n_chunks;
        // (k-steps that hold no inputs are neither fetched nor read: nothing to clear)
        fetch_chunk(0, bufp(0));
        seal();
        if (wave < kWaves / 2) {
            int bi = 0;
            for (int ch = 0; ch < nch; ch++) {
                const int bn = next(bi);
                if (ch + 1 < nch) fetch_chunk(ch + 1, bufp(bn));
                floatx16 acc;
                multiply(bufp(bi), acc);
                finish(bufp(bi), acc);
                seal();
                bi = bn;
            }
        } else {
            floatx16 acc;                                           // (chunk c - 1's sums cross barrier c in registers)
            if (1 < nch) fetch_chunk(1, bufp(1));
            multiply(bufp(0), acc);
            seal();
            int bp = 0, bi = 1;
            for (int ch = 1; ch < nch; ch++) {
                const int bn = next(bi);
                if (ch + 1 < nch) fetch_chunk(ch + 1, bufp(bn));
                finish(bufp(bp), acc);
                multiply(bufp(bi), acc);
                seal();
                bp = bi; bi = bn;
            }
            finish(bufp(bp), acc);
        }
    }
    // the two lane halves hold disjoint units of the same evaluation
    float ysum = ysum0 + ysum1;
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(ysum), __float_as_uint(ysum), false, false);
        ysum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int64_t el = (int64_t)blockIdx.x * kTl + (tid2 >> 6) * 32 + (tid2 & 31);
    if ((tid2 & 32) == 0 && el < E) {
        const int64_t ev = (int64_t)blockIdx.y * E + el;
        float y = transfer_fast(d.tf1, ysum + d.b1[0]);
        for (int q = 0; q < d.n_out_fns; q++) {
            const float *op = d.out_params + q * (1 + 2 * d.n_out);
            y = (y - op[0]) / op[1] + op[1 + d.n_out];
        }
        if (outputs) outputs[ev] = y;
        if (flags) flags[ev] = ((double)y >= d.thresholds[0]) ? 1 : 0;
    }
    (void)NE;
}

}  // namespace

// the input chains the training script writes -- [l2normalize,] one affine map -- take the chain-specialised kernel
// floats of |X| columns under one workgroup's 512 consecutive evaluations (WideDesc::front), and whether they fit behind the chunk buffers
int wide_front_stage_floats(int F, int I, int tile) { return (tile - 1) * F + I + 32 + tile + I / (F > 0 ? F : 1); }   // (+ slack, + the columns' sums of squares)
int wide_front_stage_floats(int F, int I) { return wide_front_stage_floats(F, I, kWideTile); }
bool wide_front_fits(int F, int I) { return (size_t)wide_front_stage_floats(F, I) * 4 + 2 * kChunkU4Pad * 16 <= 150 * 1024; }

bool wide_prep_is_chain(const NetDesc &n)
{
    const bool affine_last = n.n_in_fns >= 1 && n.in_fns[n.n_in_fns - 1].kind >= 3;
    return (n.n_in_fns == 2 && n.in_fns[0].kind == 0 && affine_last) || (n.n_in_fns == 1 && affine_last);
}

hipError_t launch_wide_prep(const NetDesc &n, int F, const float *columns, int C, int64_t J, int64_t E, void *xn, hipStream_t stream)
{
    if (E <= 0 || C <= 0) return hipSuccess;
    const int64_t per_block = (int64_t)(256 / kWave) * kPrepRun;
    dim3 grid((unsigned)((E + per_block - 1) / per_block), (unsigned)C);
    const bool affine_last = n.n_in_fns >= 1 && n.in_fns[n.n_in_fns - 1].kind >= 3;
    if (n.n_in_fns == 2 && n.in_fns[0].kind == 0 && affine_last)
        hipLaunchKernelGGL(wide_prep_chain_kernel<true>, grid, dim3(256), 0, stream, n, F, columns, J, E, (__bf16 *)xn);
    else if (n.n_in_fns == 1 && affine_last)
        hipLaunchKernelGGL(wide_prep_chain_kernel<false>, grid, dim3(256), 0, stream, n, F, columns, J, E, (__bf16 *)xn);
    else
        hipLaunchKernelGGL(wide_prep_kernel, grid, dim3(256), 0, stream, n, F, columns, J, E, (__bf16 *)xn);
    return hipGetLastError();
}

#ifdef SYLDET_WIDE_X_TRACE
static float *g_trace_host_ptr = nullptr;
static size_t g_trace_bytes = 0;
extern "C" int syldet_debug_wide_trace(void **ptr, size_t *bytes) { *ptr = g_trace_host_ptr; *bytes = g_trace_bytes; return 0; }
#endif

hipError_t launch_wide_gemm(const WideDesc &d, const void *xn, const float *columns, int64_t J, int64_t E, int64_t NE, float *outputs,
                            uint8_t *flags, hipStream_t stream)
{
    if (NE <= 0) return hipSuccess;
    dim3 grid((unsigned)((NE + kWideTile - 1) / kWideTile));
    // (the polynomial form exists for one output on the front-end forms only; upload_wide folded the tables for it, so anything
    // else must refuse rather than run the exp2 form on the polynomial's tables)
    if (d.poly && !(d.shape16 && d.front && d.n_out == 1)) return hipErrorInvalidValue;
    if (d.m32) {                          // the staggered two-workgroup GEMM on the 32x32x16 shape (SYLDET_WIDE_M32=1: upload_wide packed the chunks for it)
        if (!(d.front && d.n_out == 1 && !d.poly) || E <= 0 || NE % E != 0) return hipErrorInvalidValue;
        if ((size_t)wide_front_stage_floats(d.F, d.I, 256) * 4 + 2 * kChunkU4Pad * 16 > 78 * 1024) return hipErrorInvalidValue;
        const bool k19 = d.I <= 304;
        auto k32 = d.sig ? (k19 ? wide_gemm32s_kernel<true, 19> : wide_gemm32s_kernel<true, 20>) : (k19 ? wide_gemm32s_kernel<false, 19> : wide_gemm32s_kernel<false, 20>);
        const size_t lds32 = 2 * kChunkU4Pad * 16 + std::max((size_t)wide_front_stage_floats(d.F, d.I, 256) * 4, (size_t)kChunkU4Pad * 16);
        hipError_t st32 = hipFuncSetAttribute((const void *)k32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32);
        if (st32 != hipSuccess) return st32;
        hipLaunchKernelGGL(k32, dim3((unsigned)((E + 255) / 256), (unsigned)(NE / E)), dim3(512), lds32, stream, d, columns, J, E, NE, outputs, flags);
        return hipGetLastError();
    }
    if (d.shape16) {                      // the 16x16x32 shape (what ships; the other one under SYLDET_WIDE_SHAPE32=1, with its own packing)
        const bool one16 = d.n_out == 1;
        // two workgroups of 8 waves a CU (WideDesc::wg8) where the columns under 256 evaluations and the chunk buffers fit twice
        const bool wg8 = d.wg8 && d.front && one16 && (size_t)wide_front_stage_floats(d.F, d.I, 256) * 4 + 2 * kChunkU4Pad * 16 <= 78 * 1024;
        if (wg8) {
            if (d.tiles4 && d.stagger && !d.dma_builtin && (size_t)wide_front_stage_floats(d.F, d.I, 512) * 4 + 2 * kChunkU4Pad * 16 <= 158 * 1024) {
                // one workgroup of 8 waves a CU, four evaluation tiles a wave (TPW = 4)
                auto k4 = d.poly ? wide_gemm16_kernel<1, true, true, 8, true, false, 4, true>
                          : d.sig ? wide_gemm16_kernel<1, true, true, 8, true, false, 4> : wide_gemm16_kernel<1, false, true, 8, true, false, 4>;
                if (E <= 0 || NE % E != 0) return hipErrorInvalidValue;
                const size_t lds4 = 2 * kChunkU4Pad * 16 + std::max((size_t)wide_front_stage_floats(d.F, d.I, 512) * 4, (size_t)kChunkU4Pad * 16);
                hipError_t st4 = hipFuncSetAttribute((const void *)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
                if (st4 != hipSuccess) return st4;
                hipLaunchKernelGGL(k4, dim3((unsigned)((E + 511) / 512), (unsigned)(NE / E)), dim3(512), lds4, stream, d, (const uint4 *)xn, columns, J, E, NE, outputs, flags);
                return hipGetLastError();
            }
            auto k8 = d.dma_builtin ? (d.poly ? wide_gemm16_kernel<1, true, true, 8, false, true, 2, true>
                                       : d.sig ? wide_gemm16_kernel<1, true, true, 8, false, true> : wide_gemm16_kernel<1, false, true, 8, false, true>)
                      : d.stagger   ? (d.poly ? wide_gemm16_kernel<1, true, true, 8, true, false, 2, true>
                                       : d.sig ? wide_gemm16_kernel<1, true, true, 8, true> : wide_gemm16_kernel<1, false, true, 8, true>)
                                    : (d.poly ? wide_gemm16_kernel<1, true, true, 8, false, false, 2, true>
                                       : d.sig ? wide_gemm16_kernel<1, true, true, 8> : wide_gemm16_kernel<1, false, true, 8>);
            if (E <= 0 || NE % E != 0) return hipErrorInvalidValue;
            // (the staggered form's third chunk buffer lies over the stage: whichever is longer)
            const size_t lds8 = 2 * kChunkU4Pad * 16 + std::max((size_t)wide_front_stage_floats(d.F, d.I, 256) * 4, (size_t)kChunkU4Pad * 16);
            hipError_t st8 = hipFuncSetAttribute((const void *)k8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
            if (st8 != hipSuccess) return st8;
#ifdef SYLDET_WIDE_X_TRACE
            {
                const size_t need = (size_t)((E + 255) / 256) * (size_t)(NE / E) * 8 * (size_t)d.n_chunks * 5 * 1024;
                if (need > g_trace_bytes) {
                    if (g_trace_host_ptr) (void)hipFree(g_trace_host_ptr);
                    if (hipMalloc((void **)&g_trace_host_ptr, need) != hipSuccess) return hipErrorOutOfMemory;
                    g_trace_bytes = need;
                    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wide_trace), &g_trace_host_ptr, sizeof(float *));
                }
            }
#endif
            hipLaunchKernelGGL(k8, dim3((unsigned)((E + 255) / 256), (unsigned)(NE / E)), dim3(512), lds8, stream, d, (const uint4 *)xn, columns, J, E, NE, outputs, flags);
            return hipGetLastError();
        }
        auto k16 = (d.poly && d.front && one16) ? wide_gemm16_kernel<1, true, true, 16, false, false, 2, true> : d.front ? (d.sig ? (one16 ? wide_gemm16_kernel<1, true, true> : wide_gemm16_kernel<4, true, true>)
                                    : (one16 ? wide_gemm16_kernel<1, false, true> : wide_gemm16_kernel<4, false, true>))
                           : (d.sig ? (one16 ? wide_gemm16_kernel<1, true, false> : wide_gemm16_kernel<4, true, false>)
                                    : (one16 ? wide_gemm16_kernel<1, false, false> : wide_gemm16_kernel<4, false, false>));
        size_t lds16 = 2 * kChunkU4Pad * 16;
        dim3 grid16 = grid;
        if (d.front) {                    // one channel per grid row; the columns under a workgroup's evaluations behind the chunk buffers
            if (E <= 0 || NE % E != 0) return hipErrorInvalidValue;
            lds16 += (size_t)wide_front_stage_floats(d.F, d.I) * 4;
            grid16 = dim3((unsigned)((E + kWideTile - 1) / kWideTile), (unsigned)(NE / E));
        }
        hipError_t st16 = hipFuncSetAttribute((const void *)k16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
        if (st16 != hipSuccess) return st16;
        hipLaunchKernelGGL(k16, grid16, dim3(kBlock), lds16, stream, d, (const uint4 *)xn, columns, J, E, NE, outputs, flags);
        return hipGetLastError();
    }
    if (d.front) return hipErrorInvalidValue;
    // (one output: two evaluation tiles a wave; with several outputs their running sums would spill -- one tile, 16 waves)
    constexpr int TW = kWideTilesPerWave;
    const bool one = d.n_out == 1;
    auto kern = d.sig ? (one ? wide_gemm_kernel<1, true, TW> : wide_gemm_kernel<4, true, 1>)
                      : (one ? wide_gemm_kernel<1, false, TW> : wide_gemm_kernel<4, false, 1>);
    hipError_t st = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kChunkU4Pad * 16);
    if (st != hipSuccess) return st;
    hipLaunchKernelGGL(kern, grid, dim3(one ? kBlock / TW : kBlock), 2 * kChunkU4Pad * 16, stream, d, (const uint4 *)xn, NE, outputs, flags);
    return hipGetLastError();
}

}  // namespace sd
