// kernels_fused.hip -- the fast path: samples in HBM -> network outputs + detection flags in HBM in
// ONE kernel, nothing else materialised.
//
// Reference path being replaced, per frame and per evaluation (reference root relative):
//   extractPower          Common/CircularShortTimeFourierTransform.swift:280-337
//   processFourierData    Common/SyllableDetector.swift:134-151   (slice to [f0,f1))
//   processNewValue       Common/SyllableDetector.swift:153-217   (T-column window, scaling)
//   NeuralNet.apply       Common/NeuralNet.swift:294-326, :366-377
//   lastDetected          Common/SyllableDetector.swift:27-31
//
// MI355X formulation (not a translation of the vDSP call sequence):
//   * the detector only needs F (<= 32) bins of each N-point spectrum, so the windowed DFT of a
//     tile of 32 frames is the GEMM  Xt[2 x 32 rows, 32 frames] = Dt[rows, W] . S[W, 32 frames]
//     with Dt = window o {cos, -sin} and S read straight from the staged sample stream (frame j
//     is just the address j*hop: no per-frame copy, no ring).  It runs on the matrix cores as
//     v_mfma_f32_32x32x16_f16 with every operand split into f16 hi + lo (block floating point,
//     power-of-two scales): hi*hi + hi*lo + lo*hi reproduces an fp32 product to ~2^-21, and the
//     fp32 accumulate keeps the sum; measured error vs the fp64 anchor is below an fp32 FFT's.
//     The split of the samples happens on the fragment a lane has just read from LDS, in the
//     issue slots the matrix pipe leaves free (4 VALU per MFMA), so it costs no time of its own;
//   * the result tile has frames on lanes and bins in registers, which is exactly the B-operand
//     layout of the next MFMA, so the first layer -- folded with the affine input maps into
//     W' = W0 o gain, split per time slot into T*H partial dot products per frame -- runs on the
//     matrix cores too with no data movement;
//   * evaluation e is the sum over t of partial (t,h) of frame e+t.  Each partial is stored at
//     ring[h][e][t], so an evaluation reads T contiguous floats per hidden unit, plus a
//     per-window statistic (l2 norm / min-max / mean-std) -- every frame is transformed once,
//     instead of T times as in the reference's sliding re-read.
//
// Workgroup = 4 waves x 32 frames = 128 frames per pass; a workgroup walks `runs` consecutive
// passes of one channel, carrying the incomplete evaluations' partials in LDS.  HBM traffic =
// every sample once (+ (T-1) frames of overlap per segment) + 5 bytes per evaluation.
//
// The kernel is instruction-issue bound (one wave per SIMD: LDS holds 64 KB of basis fragments,
// 68 KB of samples and the ring), so the code below is written to keep the per-pass instruction
// count low: hardware-bounds-checked buffer loads, precomputed LDS offsets, no divergent guards.
//
// gfx950 only.  wave = 64.

#include "kernels.hpp"

namespace sd {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int uint32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;
constexpr int kTile = kFusedTileFrames;
// layout of the constant block in LDS (floats)
constexpr int kCstBias0 = 0, kCstRvec = 16, kCstW1 = 32, kCstB1 = 96, kCstThr = 100 /* 16 doubles */, kCstOut = 132;

__device__ __forceinline__ half8 as_half8(uint32x4 v)
{
    union { uint32x4 u; half8 h; } c;
    c.u = v;
    return c.h;
}
__device__ __forceinline__ floatx4 as_floatx4(uint32x4 v)
{
    union { uint32x4 u; floatx4 f; } c;
    c.u = v;
    return c.f;
}

// f32 pair -> packed f16 hi pair + packed f16 lo pair, hi = the top 11 significand bits (the
// conversion is then exact in any rounding mode), lo = the exact remainder rounded toward zero:
// hi + lo == x to 2^-21 relative.  6 VALU instructions per pair.
__device__ __forceinline__ void split_pair(float a, float b, unsigned &hi, unsigned &lo)
{
    const float ah = __uint_as_float(__float_as_uint(a) & 0xFFFFE000u);
    const float bh = __uint_as_float(__float_as_uint(b) & 0xFFFFE000u);
    union { decltype(__builtin_amdgcn_cvt_pkrtz(0.f, 0.f)) h; unsigned u; } ch, cl;
    ch.h = __builtin_amdgcn_cvt_pkrtz(ah, bh);
    cl.h = __builtin_amdgcn_cvt_pkrtz(a - ah, b - bh);
    hi = ch.u;
    lo = cl.u;
}

__device__ __forceinline__ void split8(floatx4 lo4, floatx4 hi4, half8 &h, half8 &l)
{
    uint32x4 uh, ul;
    unsigned a, b;
    split_pair(lo4[0], lo4[1], a, b); uh[0] = a; ul[0] = b;
    split_pair(lo4[2], lo4[3], a, b); uh[1] = a; ul[1] = b;
    split_pair(hi4[0], hi4[1], a, b); uh[2] = a; ul[2] = b;
    split_pair(hi4[2], hi4[3], a, b); uh[3] = a; ul[3] = b;
    h = as_half8(uh);
    l = as_half8(ul);
}

__device__ __forceinline__ float pow2f(int e)   // 2^e for e in [-126, 127]
{
    return __uint_as_float((unsigned)(e + 127) << 23);
}

// max(m, |x|, |y|) in one instruction (fmaxf's NaN canonicalisation costs an extra op per value)
__device__ __forceinline__ float absmax3(float m, float x, float y)
{
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(x), "v"(y), "v"(m));
    return r;
}

// Transfer functions (NeuralNet.swift:185-228).  tanh/logistic through the hardware exp2/rcp:
// absolute error below 4e-7, far inside the 1e-5 bar, at a tenth of the library call's cost.
// NaN in, NaN out (silence gives 0/0 in l2normalize; the reference then never detects).
__device__ __forceinline__ float transfer_fn(int tf, float x)
{
    if (tf == 0) {                                   // TanSig
        const float t = __builtin_amdgcn_exp2f(fminf(fabsf(x), 20.0f) * 2.885390081777927f);   // e^{2|x|}
        const float r = 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
        return x != x ? x : copysignf(r, x);
    }
    if (tf == 1) {                                   // LogSig
        const float t = __builtin_amdgcn_exp2f(fminf(fmaxf(-x, -80.0f), 80.0f) * 1.4426950408889634f);
        return x != x ? x : __builtin_amdgcn_rcpf(t + 1.0f);
    }
    if (tf == 3) return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);   // SatLin
    return x;                                        // PureLin
}

__device__ __forceinline__ floatx16 mfma16(half8 a, half8 b, floatx16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// This thread's quads of the pass whose first sample is `first` (row-relative).  The buffer
// descriptor ends one past the last sample any existing frame reads, so quads beyond it come back
// as zeros from the hardware bounds check: no per-lane guards.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float *row, int64_t first, int64_t s_eff, int nsmp)
{
    int64_t left = s_eff - first;
    left = left < 0 ? 0 : (left > nsmp ? nsmp : left);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(row + first), 0, (int)left * 4, 0x00020000);
}
template <int NL>
__device__ __forceinline__ void load_tile(const float *row, int64_t first, int64_t s_eff, int nsmp, int nload, int tid,
                                          uint32x4 (&v)[NL])
{
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(row, first, s_eff, nsmp);
#pragma unroll
    for (int k = 0; k < NL; k++)
        if (k < nload) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * tid + 4096 * k, 0, 0);
}

// Diagnostic stamps (STAMP instantiation only; never the shipped path): s_memtime at phase
// boundaries, summed per workgroup by wave 0 and stored to d.stamps[workgroup][phase].
#define SD_STAMP(slot)                                                                     \
    if (STAMP) {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        const unsigned long long now = __builtin_amdgcn_s_memtime();                       \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        tsum[slot] += now - tprev;                                                         \
        tprev = now;                                                                       \
    }

template <int KS, int MT, bool STAMP>
__global__ void __launch_bounds__(kBlock, 1)
fused_kernel(const FusedDesc d, const float *__restrict__ samples, int64_t stride, int64_t s_eff, int64_t E,
             float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32x4 *lds_dfrag = reinterpret_cast<uint32x4 *>(smem + d.lds_dfrag);
    float *smp = reinterpret_cast<float *>(smem + d.lds_hi);       // staged samples (scaled fp32)
    float *ring = reinterpret_cast<float *>(smem + d.lds_pbuf);    // [H][PS evaluations][TL]: partial (t,h) of frame e+t
    float *stat = reinterpret_cast<float *>(smem + d.lds_stat);    // [2][PS frames]: per-frame statistics
    float *red = reinterpret_cast<float *>(smem + d.lds_red);
    // evaluation-phase constants (kCst* offsets): read back with LDS latency, not a global round trip
    float *cst = reinterpret_cast<float *>(smem + d.lds_cst);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;          // frame column inside the wave's 32-frame tile
    const int hh = lane >> 5;         // lane half: k rows 8h..8h+7 of an operand, rows +4 of a result
    const int c = blockIdx.y;
    const int64_t e_b = (int64_t)blockIdx.x * d.seg_evals;
    if (e_b >= E) return;
    const int64_t e_e = (e_b + d.seg_evals < E) ? e_b + d.seg_evals : E;
    const float *row = samples + (int64_t)c * stride;
    const int PS = d.ps, T = d.T, TL = d.tl, H = d.H;
    const int fl = 32 * wave + r;     // this lane's frame inside the pass

    // DFT basis fragments -> LDS (64 KB for W = 256), once per workgroup
    for (int i = tid; i < KS * 4 * 64; i += kBlock) lds_dfrag[i] = reinterpret_cast<const uint32x4 *>(d.dfrag)[i];
    if (tid < 16) {
        cst[kCstBias0 + tid] = tid < H ? d.bias0[tid] : 0.0f;
        cst[kCstRvec + tid] = tid < H ? d.rvec[tid] : 0.0f;
        reinterpret_cast<double *>(cst + kCstThr)[tid] = tid < d.n_out ? d.thresholds[tid] : 0.0;
    }
    if (tid < 64) cst[kCstW1 + tid] = (d.n_layers == 2 && (tid >> 4) < d.n_out && (tid & 15) < H) ? d.w1[(tid >> 4) * H + (tid & 15)] : 0.0f;
    if (tid < 4) cst[kCstB1 + tid] = (d.n_layers == 2 && tid < d.n_out) ? d.b1[tid] : 0.0f;
    for (int i = tid; i < d.n_out_fns * (1 + 2 * d.n_out); i += kBlock) cst[kCstOut + i] = d.out_params[i];
    // folded first-layer fragments -> registers
    half8 wfr[MT][2][2];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int p = 0; p < 2; p++)
                wfr[m][s][p] = as_half8(reinterpret_cast<const uint32x4 *>(d.wfrag)[((m * 2 + s) * 2 + p) * 64 + lane]);
    // where partial row (m, g) of this lane's frame goes in the ring: row = h*TP + t (TP = 2^tp_log2 >= T),
    // it belongs to evaluation slot fl - t + (T-1) at position t; padding rows go to a spare word.
    int poff[MT][16];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int prow = 32 * m + (g & 3) + 8 * (g >> 2) + 4 * hh;
            const int t = prow & ((1 << d.tp_log2) - 1), h = prow >> d.tp_log2;
            poff[m][g] = (t < TL && h < H) ? ((h * PS + fl - t + (T - 1)) * TL + t) : d.ring_spare + tid;
        }
    // this lane's frame in the staged stream, and where k-step ks of lane half hh starts inside it
    const float *fptr = smp + fl * (d.hop + d.skew);
    int ko[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) ko[ks] = d.koff[ks * 2 + hh];
    const int wr0 = d.skew == 0 ? 4 * tid : 0;

    uint32x4 v[kFusedMaxLoads];
    load_tile(row, e_b * d.hop + d.gap, s_eff, d.nsmp, d.nload, tid, v);
    // results of the previous pass, stored at the start of the next one: the prefetch wait (vmcnt) then
    // never sits behind a store that was issued moments ago
    float pend_y[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int64_t pend_e = -1;
    bool pend_hit = false;
    unsigned long long tsum[16] = {0}, tprev = 0;
    if (STAMP) tprev = __builtin_amdgcn_s_memtime();

    for (int pass = 0; pass < d.runs; pass++) {
        const int64_t jp = e_b + (int64_t)kTile * pass;       // first frame of this pass
        if (jp - (T - 1) >= e_e) break;

        SD_STAMP(8)                                           // loop back-edge
        // ---------------- block floating point: scale the tile so its largest sample is in [2^13, 2^14)
        float amax = 0.0f;
#pragma unroll
        for (int k = 0; k < kFusedMaxLoads; k++)
            if (k < d.nload) {
                const floatx4 f = as_floatx4(v[k]);
                amax = absmax3(absmax3(amax, f[0], f[1]), f[2], f[3]);
            }
        SD_STAMP(9)                                           // max over the prefetched registers (vmcnt wait)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if (lane == 0) red[wave] = amax;
        SD_STAMP(0)                                           // amax (+ wait for the prefetched samples)
        __syncthreads();                                      // (A) previous pass fully consumed
        SD_STAMP(1)
        amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        int se = 13 - (int)((__float_as_uint(amax) >> 23) & 0xffu) + 127;   // exponent of the scale
        se = amax > 0.0f ? (se < -100 ? -100 : (se > 100 ? 100 : se)) : 0;
        se = __builtin_amdgcn_readfirstlane(se);
        const float sx = pow2f(se);
        if (d.skew == 0) {
#pragma unroll
            for (int k = 0; k < kFusedMaxLoads; k++)
                if (k < d.nload) {
                    const floatx4 f = as_floatx4(v[k]);
                    *reinterpret_cast<floatx4 *>(smp + wr0 + 1024 * k) = f * sx;
                }
        } else {
#pragma unroll
            for (int k = 0; k < kFusedMaxLoads; k++)
                if (k < d.nload) {
                    const int i = 4 * (tid + kBlock * k);     // quads past nsmp hold zeros and land in the buffer's slack
                    const floatx4 f = as_floatx4(v[k]);
                    *reinterpret_cast<floatx4 *>(smp + i + d.skew * (int)__umulhi((unsigned)i, d.hop_magic)) = f * sx;
                }
        }
        // evaluations the previous pass left incomplete move to the front of the ring (with their
        // frames' statistics); everything behind them starts this pass empty
        if (pass > 0) {
            // (T-1) incomplete evaluations x H rows of TL floats, copied as float2 pairs
            const int pairs = (T - 1) * (TL / 2);             // per hidden unit, contiguous in the ring
            for (int h = 0; h < H; h++)
                for (int i = tid; i < pairs; i += kBlock) {
                    floatx2 *base = reinterpret_cast<floatx2 *>(ring + h * PS * TL);
                    base[i] = base[kTile * (TL / 2) + i];
                }
            if (tid < T - 1) {
                stat[tid] = stat[kTile + tid];
                stat[PS + tid] = stat[PS + kTile + tid];
            }
        }
        SD_STAMP(2)                                           // scale + LDS stage writes + carry
        __syncthreads();                                      // (B) samples staged
        SD_STAMP(3)

        SD_STAMP(10)
        // previous pass's results out (2-layer networks; see the evaluation phase)
        if (pend_e >= 0) {
            if (outputs) {
#pragma unroll
                for (int o = 0; o < 4; o++)
                    if (o < d.n_out) outputs[((int64_t)c * E + pend_e) * d.n_out + o] = pend_y[o];
            }
            if (flags) flags[(int64_t)c * E + pend_e] = pend_hit ? 1 : 0;
            pend_e = -1;
        }
        // next pass's samples: fetched during this pass's matrix work, a load or two per k-step, so the
        // memory queue never backs up into the wave (a burst of 17 KB-sized loads stalls issue for ~3k cycles)
        const __amdgpu_buffer_rsrc_t nrs = tile_rsrc(row, (jp + kTile) * d.hop + d.gap, pass + 1 < d.runs ? s_eff : 0, d.nsmp);
        constexpr int kLoadsPerStep = (kFusedMaxLoads + KS - 1) / KS;
        SD_STAMP(11)                                          // deferred stores + prefetch issue

        // ---------------- band-limited DFT of this wave's 32 frames on the matrix cores.
        // Software pipeline: while the six MFMAs of k-step ks execute, the lane's next 8 samples
        // (already in registers) are split into f16 hi/lo and the fragments after that are fetched.
        floatx16 acc_re = {0}, acc_im = {0};
        floatx4 s0 = *reinterpret_cast<const floatx4 *>(fptr + ko[0]);
        floatx4 s1 = *reinterpret_cast<const floatx4 *>(fptr + ko[0] + 4);
        half8 bh, bl;
        split8(s0, s1, bh, bl);
        if (KS > 1) {
            s0 = *reinterpret_cast<const floatx4 *>(fptr + ko[1]);
            s1 = *reinterpret_cast<const floatx4 *>(fptr + ko[1] + 4);
        }
        uint32x4 a0 = lds_dfrag[0 * 64 + lane], a1 = lds_dfrag[1 * 64 + lane];
        uint32x4 a2 = lds_dfrag[2 * 64 + lane], a3 = lds_dfrag[3 * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);                    // the prologue's fetches stay out of the loop's groups
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const half8 a_re_h = as_half8(a0), a_re_l = as_half8(a1), a_im_h = as_half8(a2), a_im_l = as_half8(a3);
            const half8 cbh = bh, cbl = bl;
            const floatx4 n0 = s0, n1 = s1;
            if (ks + 1 < KS) {                                // fragments of the next k-step
                a0 = lds_dfrag[((ks + 1) * 4 + 0) * 64 + lane];
                a1 = lds_dfrag[((ks + 1) * 4 + 1) * 64 + lane];
                a2 = lds_dfrag[((ks + 1) * 4 + 2) * 64 + lane];
                a3 = lds_dfrag[((ks + 1) * 4 + 3) * 64 + lane];
            }
            if (ks + 2 < KS) {                                // raw samples two k-steps ahead
                s0 = *reinterpret_cast<const floatx4 *>(fptr + ko[ks + 2]);
                s1 = *reinterpret_cast<const floatx4 *>(fptr + ko[ks + 2] + 4);
            }
#pragma unroll
            for (int j = 0; j < kLoadsPerStep; j++) {         // next pass's quads ks*kLoadsPerStep + j
                const int k = ks * kLoadsPerStep + j;
                if (k < kFusedMaxLoads && k < d.nload) v[k] = __builtin_amdgcn_raw_buffer_load_b128(nrs, 16 * tid + 4096 * k, 0, 0);
            }
            acc_re = mfma16(a_re_h, cbh, acc_re);
            acc_im = mfma16(a_im_h, cbh, acc_im);
            acc_re = mfma16(a_re_h, cbl, acc_re);
            acc_im = mfma16(a_im_h, cbl, acc_im);
            acc_re = mfma16(a_re_l, cbh, acc_re);
            acc_im = mfma16(a_im_l, cbh, acc_im);
            if (ks + 1 < KS) split8(n0, n1, bh, bl);          // 24 VALU, scheduled into the MFMA shadows below
            // schedule: this k-step's 6 LDS fetches first (their data is used one and two k-steps later),
            // then each MFMA followed by 4 of the split's VALU instructions
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);       // 6 DS reads
#pragma unroll
            for (int i = 0; i < 6; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // 4 VALU
            }
        }

        SD_STAMP(4)                                           // DFT MFMA loop
        // ---------------- magnitude (zvabs/2 :329-333 or zvmags/4 :270-274), scaling
        // (SyllableDetector.swift:184-212), per-frame statistic, f16 split for the next MFMA.
        // Result layout: column = frame r, register g of lane half hh = bin row (g&3) + 8(g>>2) + 4hh.
        const float inv = pow2f(-se - 13);                   // accumulators hold X * sx * 2^13
        float cs = 1.0f;                                     // keeps the column inside f16 range (power of two)
        if (d.scaling == 0) cs = d.power_mode ? pow2f(2 * (se < 40 ? (se > -40 ? se : -40) : 40) - 30) : pow2f(se - 8);
        const int fh = d.F - 4 * hh;                          // register g holds a band row iff (g&3) + 8(g>>2) < fh
        float cval[16];
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const float re = acc_re[g] * inv, im = acc_im[g] * inv;
            const float pw = fmaf(re, re, im * im);
            cval[g] = d.power_mode ? pw : __builtin_amdgcn_sqrtf(pw);
        }
        if (d.scaling != 0) {
            const float k = d.scaling == 1 ? 0.6931471805599453f : 6.020599913279624f;   // ln 2, 20 log10 2
#pragma unroll
            for (int g = 0; g < 16; g++) cval[g] = k * __builtin_amdgcn_logf(cval[g]);   // v_log_f32 = log2
        }
        if (d.F < 32 || d.scaling != 0) {
#pragma unroll
            for (int g = 0; g < 16; g++) cval[g] = ((g & 3) + 8 * (g >> 2)) < fh ? cval[g] : 0.0f;
        }
        const int slot = (T - 1) + fl;
        if (d.norm == 1) {
            float st0 = 0.0f;
#pragma unroll
            for (int g = 0; g < 16; g++) st0 = fmaf(cval[g], cval[g], st0);
            st0 += __shfl_xor(st0, 32, 64);
            if (hh == 0) stat[slot] = st0;
        } else if (d.norm == 2) {
            float st0 = INFINITY, st1 = -INFINITY;
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const bool valid = ((g & 3) + 8 * (g >> 2)) < fh;
                st0 = valid ? fminf(st0, cval[g]) : st0;
                st1 = valid ? fmaxf(st1, cval[g]) : st1;
            }
            st0 = fminf(st0, __shfl_xor(st0, 32, 64));
            st1 = fmaxf(st1, __shfl_xor(st1, 32, 64));
            if (hh == 0) { stat[slot] = st0; stat[PS + slot] = st1; }
        } else if (d.norm == 3) {
            float st0 = 0.0f, st1 = 0.0f;
#pragma unroll
            for (int g = 0; g < 16; g++) st0 += cval[g];
            st0 += __shfl_xor(st0, 32, 64);
            st0 = st0 / (float)d.F;                           // mean of this frame's column
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const float dlt = cval[g] - st0;
                st1 = ((g & 3) + 8 * (g >> 2)) < fh ? fmaf(dlt, dlt, st1) : st1;   // M2 of this frame's column
            }
            st1 += __shfl_xor(st1, 32, 64);
            if (hh == 0) { stat[slot] = st0; stat[PS + slot] = st1; }
        }
        half8 bh2[2], bl2[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            floatx4 lo4, hi4;
#pragma unroll
            for (int j = 0; j < 4; j++) { lo4[j] = cval[8 * s + j] * cs; hi4[j] = cval[8 * s + 4 + j] * cs; }
            split8(lo4, hi4, bh2[s], bl2[s]);
        }

        // ---------------- first layer, folded and split per time slot: P[(t,h)][frame] on the matrix cores
        const float unscale = d.w_unscale / cs;
#pragma unroll
        for (int m = 0; m < MT; m++) {
            floatx16 pacc = {0};
#pragma unroll
            for (int s = 0; s < 2; s++) {
                pacc = mfma16(wfr[m][s][0], bh2[s], pacc);
                pacc = mfma16(wfr[m][s][0], bl2[s], pacc);
                pacc = mfma16(wfr[m][s][1], bh2[s], pacc);
            }
#pragma unroll
            for (int g = 0; g < 16; g++) ring[poff[m][g]] = pacc[g] * unscale;
        }
        SD_STAMP(5)                                           // magnitude, statistic, layer-0 MFMA, ring writes
        __syncthreads();                                      // (C) partials of all 128 frames visible
        SD_STAMP(6)

        // ---------------- evaluations completed by this pass: slot q = e - (jp - (T-1)); lane half hh
        // owns hidden units hh, hh+2, ...; both halves hold the same evaluation
        {
            const int q = fl;
            const int64_t e = jp - (T - 1) + q;
            const bool valid = e >= e_b && e < e_e;
            float alpha = 1.0f, beta = 0.0f;
            if (d.norm == 1) {                                // L2Normalize, NeuralNet.swift:47-59
                float ssw = 0.0f;
#pragma unroll 4
                for (int t = 0; t < T; t++) ssw += stat[q + t];
                alpha = __builtin_amdgcn_rsqf(ssw);
            } else if (d.norm == 2) {                         // Normalize, :69-96
                float mn = INFINITY, mx = -INFINITY;
                for (int t = 0; t < T; t++) { mn = fminf(mn, stat[q + t]); mx = fmaxf(mx, stat[PS + q + t]); }
                const float range = mx - mn;
                if (range == 0.0f) { alpha = 0.0f; beta = -1.0f; }
                else { alpha = 2.0f / range; beta = (0.0f - mn - mx) / range; }
            } else if (d.norm == 3) {                         // NormalizeStd, :105-108 (population sigma)
                float n = 0.0f, mean = 0.0f, m2 = 0.0f;
                for (int t = 0; t < T; t++) {                 // pairwise-stable combination of per-frame (mean, M2)
                    const float nb = (float)d.F, tot = n + nb, dlt = stat[q + t] - mean;
                    mean += dlt * nb / tot;
                    m2 += stat[PS + q + t] + dlt * dlt * n * nb / tot;
                    n = tot;
                }
                const float sd = sqrtf(m2 / (float)d.I);
                alpha = 1.0f / sd;
                beta = -mean / sd;
            }
            const int64_t obase = ((int64_t)c * E + e) * d.n_out;
            const double *thr = reinterpret_cast<const double *>(cst + kCstThr);
            float y[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            bool hit = false;
            for (int h = hh; h < H; h += 2) {
                const floatx2 *pp = reinterpret_cast<const floatx2 *>(ring + (h * PS + q) * TL);
                const float b0 = cst[kCstBias0 + h], rv = cst[kCstRvec + h];
                const float w10 = cst[kCstW1 + h], w11 = cst[kCstW1 + 16 + h], w12 = cst[kCstW1 + 32 + h], w13 = cst[kCstW1 + 48 + h];
                float z0 = 0.0f, z1 = 0.0f;
#pragma unroll 8
                for (int t = 0; t < TL / 2; t++) { const floatx2 p2 = pp[t]; z0 += p2[0]; z1 += p2[1]; }
                float a = transfer_fn(d.tf0, fmaf(alpha, z0 + z1, fmaf(beta, rv, b0)));
                if (d.n_layers == 2) {
                    y[0] = fmaf(w10, a, y[0]); y[1] = fmaf(w11, a, y[1]); y[2] = fmaf(w12, a, y[2]); y[3] = fmaf(w13, a, y[3]);
                } else {
                    for (int k = 0; k < d.n_out_fns; k++) {                    // reverse maps, NeuralNet.swift:137-142 / :175-180
                        const float *op = cst + kCstOut + k * (1 + 2 * d.n_out);
                        a = (a - op[0]) / op[1 + h] + op[1 + d.n_out + h];
                    }
                    if (valid && outputs) outputs[obase + h] = a;
                    if (h == 0 || d.rule == 1) hit = hit || ((double)a >= thr[h]);
                }
            }
            if (d.n_layers == 2) {
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    if (o < d.n_out) {
                        float yo = y[o] + __shfl_xor(y[o], 32, 64);
                        yo = transfer_fn(d.tf1, yo + cst[kCstB1 + o]);
                        for (int k = 0; k < d.n_out_fns; k++) {
                            const float *op = cst + kCstOut + k * (1 + 2 * d.n_out);
                            yo = (yo - op[0]) / op[1 + o] + op[1 + d.n_out + o];
                        }
                        pend_y[o] = yo;
                        if (o == 0 || d.rule == 1) hit = hit || ((double)yo >= thr[o]);
                    }
                }
                pend_hit = hit;
                pend_e = (valid && hh == 0) ? e : -1;
            } else {
                hit = hit || (__shfl_xor((int)hit, 32, 64) != 0);
                if (valid && hh == 0 && flags) flags[(int64_t)c * E + e] = hit ? 1 : 0;
            }
        }
        SD_STAMP(7)                                           // evaluations
    }
    if (pend_e >= 0) {
        if (outputs) {
#pragma unroll
            for (int o = 0; o < 4; o++)
                if (o < d.n_out) outputs[((int64_t)c * E + pend_e) * d.n_out + o] = pend_y[o];
        }
        if (flags) flags[(int64_t)c * E + pend_e] = pend_hit ? 1 : 0;
    }
    if (STAMP && tid == 0 && d.stamps)
        for (int i = 0; i < 16; i++) d.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + i] = tsum[i];
}

template <int KS, int MT, bool STAMP = false>
hipError_t launch_one(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t s_eff, int64_t E,
                      float *outputs, uint8_t *flags, hipStream_t stream)
{
    auto kern = fused_kernel<KS, MT, STAMP>;
    hipError_t st = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, d.lds_total);
    if (st != hipSuccess) return st;
    const int64_t segs = (E + d.seg_evals - 1) / d.seg_evals;
    dim3 grid((unsigned)segs, (unsigned)C);
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), (size_t)d.lds_total, stream, d, samples, stride, s_eff, E, outputs, flags);
    return hipGetLastError();
}

}  // namespace

int fused_supported(int KS, int MT)
{
    return (KS == 16 || KS == 8) && (MT == 1 || MT == 2 || MT == 4) ? 1 : 0;
}

hipError_t launch_fused(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t S, int64_t J,
                        int64_t E, float *outputs, uint8_t *flags, hipStream_t stream)
{
    (void)S;
    if (E <= 0 || C <= 0) return hipSuccess;
    // one past the last sample an existing frame reads: frame J-1 covers [(J-1)*hop + gap, ... + W)
    const int64_t s_eff = (J - 1) * (int64_t)d.hop + d.gap + d.W;
    if (d.stamps && d.KS == 16 && d.MT == 2) return launch_one<16, 2, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
#define SD_CASE(K, M) \
    if (d.KS == K && d.MT == M) return launch_one<K, M>(d, samples, stride, C, s_eff, E, outputs, flags, stream)
    SD_CASE(16, 1);
    SD_CASE(16, 2);
    SD_CASE(16, 4);
    SD_CASE(8, 1);
    SD_CASE(8, 2);
    SD_CASE(8, 4);
#undef SD_CASE
    return hipErrorInvalidValue;
}

}  // namespace sd
