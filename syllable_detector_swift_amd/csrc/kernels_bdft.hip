// kernels_bdft.hip -- long frames whose hop divides them (BASELINE configs[2]: 1024-point frames, hop 256): samples in HBM ->
// network outputs + detection flags in HBM in ONE kernel, with the transform on the matrix cores and every sample
// transformed ONCE, not once per frame that covers it.
//
// Reference path being replaced, per frame and per evaluation (reference root relative):
//   extractPower          Common/CircularShortTimeFourierTransform.swift:280-337   (window, packed real FFT, |X|)
//   processFourierData    Common/SyllableDetector.swift:134-151                    (slice to [f0, f1))
//   processNewValue       Common/SyllableDetector.swift:153-217                    (timeRange-column window)
//   NeuralNet.apply       Common/NeuralNet.swift:294-326, :366-377                 (l2normalize, affine maps, TanSig, linear)
//   lastDetected          Common/SyllableDetector.swift:27-31
//
// With W = N = R hop a frame is R consecutive blocks of `hop` samples, and its DFT under a rectangular window is
//     Y_j[k] = sum_{q < R} w_R^{k q} B_{j+q}[k],   w_R = e^{-2 pi i / R},   B_b[k] = sum_{n < hop} x[b hop + n] e^{-2 pi i k n / N}
// -- a block's band-limited partial transform is shared by the R frames that contain it, and for R = 4 the factors are
// 1, -i, -1, i: a sum of four with swaps and signs.  The reference's windows are short cosine sums, w[n] = sum_p a_p
// cos(2 pi p n / N) (Hamming 0.54, -0.46; Hann 0.5, -0.5; rectangular 1), so the window is three taps along the bins AFTER the
// transform:  X_j[k] = a_0 Y_j[k] + (a_1 / 2) (Y_j[k-1] + Y_j[k+1]).  B_b itself is kernels_fused_s.hip's symmetric fold (a block
// under no window is symmetric about its centre c = hop / 2): with s[m] = x[c+m] + x[c-m], d[m] = x[c+m] - x[c-m],
//     B'_b[k] = e^{+i theta k} B_b[k] = sum_m s[m] cos(2 pi k m / N) - i sum_m d[m] sin(2 pi k m / N) + x[0] e^{+i theta k},  theta = 2 pi c / N,
// two GEMMs of K = hop / 2 on the matrix cores (f16 hi + lo operands, three products, fp32 accumulation).  The common phase
// e^{i theta k} drops out of |X| once the window's taps carry e^{+-i theta}.
// Per frame: 12 matrix instructions instead of the FFT kernel's ~280 vector instructions and 18 KB of LDS transposes.
//
// A workgroup of 8 waves (two per SIMD) walks a contiguous run of one channel.  Wave w owns bins kb0 + 16 w .. + 15: their
// cosine rows and sine rows are 64 registers, resident.  The run advances in sub-tiles of 16 blocks with ONE barrier each:
//     iteration u:  barrier  |  load the raw samples of sub-tile u+2 into registers (each of 512 threads: 8 positions of one block)
//                            |  fold, scale (the block's own power of two), split sub-tile u+1 -> B fragments in LDS (the other buffer)
//                            |  24 MFMAs on sub-tile u's fragments, the first sample's rank-1 term, back to true units,
//                               the sliding sum over the last R blocks (DPP row shifts; the previous sub-tile's last columns carry)
//                            |  the lane groups' edge bins of sub-tile u -> LDS
//                            |  window taps, |X|, f16 hi + lo columns of sub-tile u-1 (its neighbours' edge bins arrived with the barrier)
//                            |  block maxima of sub-tile u+2 (LDS atomic max)
// Six sub-tiles (96 new frames) make a tile; its end is kernels_fft1k.hip's back half: all taps of the first layer as the rows of one
// GEMM over the tile's new columns, four threads per evaluation for the rest of the network, the last timeRange - 1 columns carried
// in a ring of 112 column rows.  Since round 5 the stream of sub-tiles does not stop for it: tile N's last columns are finished in
// tile N + 1's first iteration, its tap products made in the second, its evaluations in the third (stage_taps, stage_evaluate).
// What the ISA must look like for the speed measured (MEASUREMENTS R5.4): built without the SLP vectoriser, the sliding
// sums' lane shifts inside their additions (tools/check_dpp_fusion.py checks it at build time).
//
// gfx950 only.  wave = 64.

#include <algorithm>
#include <cstdio>
#include <type_traits>

#include "fused_common.hpp"

namespace sd {

#ifdef SYLDET_B_STAMPS                  // diagnostic build only (tools/knockouts.sh): cycles per stage, summed over workgroups, printed by the launcher
__device__ unsigned long long g_bdft_stamps[16];
#endif

namespace {

using namespace fused_dev;

constexpr int kBlock = kBdftBlock;             // 512 threads = 8 waves, two per SIMD
constexpr int kWaves = kBlock / 64;
constexpr int kTile = 112;                     // rows of the column ring: timeRange - 1 carried + 96 new fit (seven row tiles of 16; not a power of two: wrap())
#ifndef SYLDET_B_PS
#define SYLDET_B_PS 52                        // (diagnostic builds try other strides)
#endif
// floats per row of tap products: 12 taps x 4 units and a slot of padding.  13 slots of 16 bytes a row: the tap stage's 16-byte
// stores (eight consecutive lanes = eight consecutive rows at a time, 32 banks) fall on eight different slots; the evaluations'
// 16-byte reads (sixteen lanes at a time in the hardware's own grouping, 64 banks: MI355X_MICROARCH.md, LDS) do too once the sixteen
// evaluations of a wave are dealt to its quads in the order kEvalOrder -- with rows 12 slots apart the stores were four-way
// conflicts (a third of the kernel's bank-conflict cycles, tools/bdft_conflicts.sh), with 13 and the plain order the reads two-way
constexpr int kPS = SYLDET_B_PS;
constexpr unsigned long long kEvalOrder = 0xfd57ce64b9138a20ull;       // quad q of a wave takes its evaluation (kEvalOrder >> 4 q) & 15
__device__ __forceinline__ int wrap(int r) { return r >= kTile ? r - kTile : r; }      // (r < 2 kTile)
constexpr int kSubs = 6;                       // sub-tiles of 16 blocks per tile
constexpr int kNew = 16 * kSubs;               // new frames (= evaluations) per tile

__device__ __forceinline__ float rem_lo(float a, unsigned h)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(h));
    return r;
}
__device__ __forceinline__ float rem_hi(float a, unsigned h)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(h));
    return r;
}
__device__ __forceinline__ unsigned cvt_pk(float a, float b)
{
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo)
{
    hi = cvt_pk(a, b);
    lo = cvt_pk(rem_lo(a, hi), rem_hi(b, hi));
}
// the value of lane n - Q of this 16-lane row (0 where there is none) / of lane n + 16 - Q of `prev` (0 elsewhere)
template <int Q>
__device__ __forceinline__ float shr_cur(float v)
{
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x110 + Q, 0xF, 0xF, true));   // row_shr:Q
}
template <int Q>
__device__ __forceinline__ float shl_prev(float v)
{
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x100 + (16 - Q), 0xF, 0xF, true));   // row_shl:16-Q
}
// lead + column n - Q of the stream.  One of the two shifted values is zero in every lane, so the order of the additions does not
// change a bit of the result -- but in this order each shift folds into its addition (v_add_f32_dpp): two instructions, not four
template <int Q>
__device__ __forceinline__ float back_add(float lead, float cur, float prev) { return (lead + shr_cur<Q>(cur)) + shl_prev<Q>(prev); }

// KS: k-steps of 32 folded positions per block (hop = 64 KS).  RR: blocks per frame (4, 2, 1).  SC: log / dB columns.
template <int KS, bool SC, int RR>
__global__ void __launch_bounds__(kBlock, 1)
bdft_net_kernel(const MlpxDesc d, const BdftDesc bd, const float *__restrict__ samples, int64_t stride, int64_t S, int64_t J, int64_t E,
                int64_t evals_per_run, float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KB = 4, R = RR, HOP = 64 * KS;
    // LDS: first-layer fragments | columns hi | columns lo | per-frame sums, exponents | B fragments (two buffers) | edge bins |
    // per-wave partial sums | block maxima | records | the tap products of a tile.  (Round 5: the products have a place of their own --
    // they used to sit where the second fragment buffer and the edge bins are -- so that a tile's end can run under the next tile's
    // first iterations; the room came from a ring of 112 column rows instead of 128: 158 of the CU's 160 KB.)
    const int CS = d.col_stride;
    constexpr int PS = kPS;
    uint32x4 *afr = reinterpret_cast<uint32x4 *>(smem);
    _Float16 *colh = reinterpret_cast<_Float16 *>(smem + 3 * KB * 2 * 1024);
    _Float16 *coll = colh + kTile * CS;
    float *ssf = reinterpret_cast<float *>(coll + kTile * CS);       // [tile] per-frame sums of squares (true units)
    float *fsc = ssf + kTile;                                        // [tile] a frame's products back to true units
    unsigned char *scr = reinterpret_cast<unsigned char *>(fsc + kTile);
    constexpr int kFragBytes = 4 * KS * 1024 < 16384 ? 16384 : 4 * KS * 1024;   // s hi, s lo, d hi, d lo: KS fragments each (and room for the products)
    uint32x2 *bfr0 = reinterpret_cast<uint32x2 *>(scr), *bfr1 = reinterpret_cast<uint32x2 *>(scr + kFragBytes);
    floatx4 *edges = reinterpret_cast<floatx4 *>(scr + 2 * kFragBytes);                  // [2 parities][16 frames][32 lane groups] (re0, im0, re3, im3)
    float *ssf8 = reinterpret_cast<float *>(scr + 2 * kFragBytes + 2 * 16 * 32 * 16);   // [tile][8 waves]
    unsigned *bmax = reinterpret_cast<unsigned *>(ssf8 + kTile * kWaves);                 // [4][16] block maxima (bit patterns of |x|), by sub-tile & 3
    // [4][16] what the multiplying waves need of a block and of the frame ending on it, made once by the fold:
    // (x[0] 2^e, 2^(-e-13): the block's first sample in the fragments' units and the way back; the frame's column scale up, down)
    floatx4 *recs = reinterpret_cast<floatx4 *>(bmax + 64);
    float *pbuf = reinterpret_cast<float *>(recs + 64);                                  // [ring rows][PS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int c = blockIdx.y;
    const int F = d.F, T = d.T;
    const int scaling = SC ? d.scaling : 0;                          // SYLDET_SCALING_*: linear |X|, ln |X| or 20 log10 |X| columns
    const float lscale = scaling == 1 ? 0.6931471805599453f : 6.020599913279624f;      // ln 2, 20 log10 2
    const int64_t E0 = (int64_t)blockIdx.x * evals_per_run;
    if (E0 >= E) return;
    const int64_t E1 = E0 + evals_per_run < E ? E0 + evals_per_run : E;
    const int64_t fbase = E0 - 16;                                   // the run's first (discarded) frame: a sub-tile of lead-in
    const int tiles = (int)((E1 - E0 + 16 + (T - 1) + kNew - 1) / kNew);

    // ---- once per workgroup: first-layer fragments, zero columns (rows the run never writes must not hold NaNs), maxima
    for (int i = tid; i < 3 * KB * 2 * 64; i += kBlock) afr[i] = reinterpret_cast<const uint32x4 *>(bd.afrag)[i];
    for (int i = tid; i < kTile * CS / 2; i += kBlock) {
        reinterpret_cast<unsigned *>(colh)[i] = 0u;
        reinterpret_cast<unsigned *>(coll)[i] = 0u;
    }
    if (tid < kTile) { ssf[tid] = 0.0f; fsc[tid] = 0.0f; }
    for (int i = tid; i < kTile * kWaves; i += kBlock) ssf8[i] = 0.0f;
    for (int i = tid; i < kTile * kPS; i += kBlock) pbuf[i] = 0.0f;                         // (the first tile's carried rows: no frames, no products)
    if (tid < 64) bmax[tid] = 0u;
    const float b0[4] = {d.bias0[0], d.bias0[1], d.bias0[2], d.bias0[3]}, w1[4] = {d.w1[0], d.w1[1], d.w1[2], d.w1[3]};
    const double thr = d.thresholds[0];
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(outputs ? outputs + (int64_t)c * E : nullptr, 0, outputs ? (int)(E * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t flg_rs = __builtin_amdgcn_make_buffer_rsrc(flags ? flags + (int64_t)c * E : nullptr, 0, flags ? (int)E : 0, 0x00020000);
    const float *chan = samples + (int64_t)c * stride;
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(chan), 0, (int)(S * 4), 0x00020000);

    // ---- this wave's basis: cosine rows and sine rows of its 16 bins, [k-step][hi, lo]; the first sample's real coefficients
    half8 ac[KS][2], as_[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int p = 0; p < 2; p++) {
            ac[ks][p] = as_half8(reinterpret_cast<const uint32x4 *>(bd.basis)[(((wave * 2 + 0) * KS + ks) * 2 + p) * 64 + lane]);
            as_[ks][p] = as_half8(reinterpret_cast<const uint32x4 *>(bd.basis)[(((wave * 2 + 1) * KS + ks) * 2 + p) * 64 + lane]);
        }
    float cre[4];
#pragma unroll
    for (int i = 0; i < 4; i++) cre[i] = bd.cre[(wave * 64 + lane) * 4 + i];
    const float wa0 = bd.a0, wc = bd.a1c, wsn = bd.a1s;               // a_0, (a_1 / 2) cos theta, (a_1 / 2) sin theta
    const int fb0 = bd.kb0 + 16 * wave + 4 * g - bd.f0;              // band index of this lane's first bin (the band is [0, F))
    const int lane_col = n * CS + 16 * wave + 4 * g, lane_ssf = n * kWaves + wave;
    // the taps per bin: zero outside the band where the columns are linear -- such a column is then 0 by itself, its square needs no
    // mask in the frame's sum (log columns keep the mask: a logarithm is not zero there)
    float wA[4], wC[4], wS[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const bool on = SC || (unsigned)(fb0 + i) < (unsigned)F;
        wA[i] = on ? wa0 : 0.0f; wC[i] = on ? wc : 0.0f; wS[i] = on ? wsn : 0.0f;
#ifdef SYLDET_B_TAPS_UNIFORM
        wA[i] = wa0; wC[i] = wc; wS[i] = wsn;
#endif
    }

    // ---- the fold's thread layout: block n of the sub-tile, positions m = 32 ks + 8 gf + 4 half + j (j < 4)
    // (which half of its eight positions a thread folds alternates with bit 3 of its block: an 8-byte store is served sixteen
    // consecutive lanes at a time over 32 banks (MI355X_MICROARCH.md, LDS), and with one half per WAVE blocks n and n + 8 -- 16-byte
    // slots 128 bytes apart -- met in the same banks: 39 % of the kernel's bank-conflict cycles, tools/bdft_conflicts.sh)
    const int f_ks = wave % KS, f_half = ((wave / KS) ^ (n >> 3)) & 1;
    const bool folder = wave < 2 * KS;
    const int f_m = 32 * f_ks + 8 * g + 4 * f_half;
    // raw samples of one block's share: x[c + m .. c + m + 3], x[c - m - 4 .. c - m - 1], x[c - m], and the block's first sample
    struct Raw { floatx4 p, q; float z, x0; };
    auto load_raw = [&](int64_t blk0) {                              // blk0: first block of the sub-tile
        Raw r;
        const int base = (int)(blk0 + n) * HOP;                       // (in samples; S 4 < 2^31, negative in the lead-in)
        const bool ok = base >= 0 && folder;
        const unsigned o = (unsigned)(base + HOP / 2) * 4u;
        r.p = as_floatx4(__builtin_amdgcn_raw_buffer_load_b128(in_rs, ok ? o + (unsigned)f_m * 4u : 0xFFFFFFF0u, 0, 0));
        r.q = as_floatx4(__builtin_amdgcn_raw_buffer_load_b128(in_rs, ok ? o - (unsigned)(f_m + 4) * 4u : 0xFFFFFFF0u, 0, 0));
        r.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(in_rs, ok ? o - (unsigned)f_m * 4u : 0xFFFFFFF0u, 0, 0));
        r.x0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(in_rs, (ok && (wave == 0 || f_m == 0)) ? (unsigned)base * 4u : 0xFFFFFFF0u, 0, 0));   // (wave 0: the block's record; f_m == 0: slot 0 of the differences)
        return r;
    };
    auto raw_max = [&](const Raw &r, unsigned *slot) {               // this thread's share of its block's loudest sample
        float m = absmax3(absmax3(0.0f, r.p[0], r.p[1]), r.p[2], r.p[3]);
        m = absmax3(absmax3(m, r.q[0], r.q[1]), r.q[2], r.q[3]);
        m = absmax3(m, r.z, r.x0);
        // the four lanes of this wave that share a block first (two swaps), then one LDS atomic a block and wave instead of four
        unsigned mu = __float_as_uint(m);                            // (v_max3 drops NaNs: non-negative numbers compare as their bits)
        auto rr = __builtin_amdgcn_permlane16_swap(mu, mu, false, false);
        mu = max(rr[0], rr[1]);
        rr = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
        mu = max(rr[0], rr[1]);
        if (folder && g == 0) atomicMax(slot + n, mu);
    };
    auto scale_exp = [](unsigned bits) {                             // 2^e puts the block's loudest sample into [2^13, 2^14)
        const int ex = (int)((bits >> 23) & 0xffu);
        int e = 140 - ex;
        e = bits != 0u ? (e < -100 ? -100 : (e > 113 ? 113 : e)) : 0;
        return e;
    };
    auto fold_store = [&](const Raw &r, int slt, uint32x2 *bf, bool writers) {   // slt: the sub-tile's slot (its index & 3); writers: waves 0 and 1 may be here
        if (!folder) return;
        const unsigned *slot = bmax + 16 * slt;
        const int eb = scale_exp(slot[n]);
        const float sx = pow2f(eb);
        if (writers && wave == 0 && g == 0) *reinterpret_cast<floatx2 *>(recs + 16 * slt + n) = floatx2{r.x0 * sx, pow2f(-eb - 13)};
        if (writers && wave == 1 && g == 0) {                        // the frame ending on block n: the loudest of its four blocks (silent ones do not count)
            const unsigned *prevs = bmax + 16 * ((slt + 3) & 3);      // (the sub-tile before this one)
            int e = 0x7fff;
#pragma unroll
            for (int q = 0; q < R; q++) {
                const unsigned mbq = n - q >= 0 ? slot[n - q] : prevs[16 + n - q];
                e = mbq != 0u ? min(e, scale_exp(mbq)) : e;
            }
            // the frame's column scale (every wave reads the same power of two: |X| 2^(e - 10) < 2^14)
            const int ef = e == 0x7fff ? 0 : e;
            const int eu = ef - 10 < -120 ? -120 : (ef - 10 > 120 ? 120 : ef - 10);
            *(reinterpret_cast<floatx2 *>(recs + 16 * slt + n) + 1) = floatx2{pow2f(eu), pow2f(-eu)};
        }
        const float xp[4] = {r.p[0], r.p[1], r.p[2], r.p[3]};
        const float xm[4] = {r.z, r.q[3], r.q[2], r.q[1]};           // x[c - m - j]
        float s[4], dd[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float tm = xm[j] * sx;
            s[j] = fmaf(xp[j], sx, tm);
            dd[j] = fmaf(xp[j], sx, -tm);
        }
        if (f_m == 0) dd[0] = r.x0 * sx;                             // slot 0 of the differences carries the block's first sample
        uint32x2 sh, sl, dh, dl;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            unsigned h, l;
            split2(s[2 * j], s[2 * j + 1], h, l);
            sh[j] = h; sl[j] = l;
            split2(dd[2 * j], dd[2 * j + 1], h, l);
            dh[j] = h; dl[j] = l;
        }
        // fragment [which][k-step][lane (n, g)]: 8 halves; this thread's four are its half of them
        uint32x2 *at = bf + ((f_ks * 64 + 16 * g + n) * 2 + f_half);
        at[0 * KS * 128] = sh;
        at[1 * KS * 128] = sl;
        at[2 * KS * 128] = dh;
        at[3 * KS * 128] = dl;
    };

    // ---- the stream's state
    floatx4 yre = {0.f, 0.f, 0.f, 0.f}, yim = yre;                   // the last multiplied sub-tile: Y' of this lane's four bins for the frame ending at block n
    floatx4 bre_prev = yre, bim_prev = yre;                          // ... the blocks' own partial transforms (true units), for the carry
    floatx4 vre_prev = yre, vim_prev = yre;                          // ... and the first level of the sliding sum
    float upc = 0.0f, dnc = 0.0f;                                    // ... and the frames' column scales
    int64_t u = 0;                                                   // sub-tile counter of the run
    // the ring of column rows: sub-tile q's sixteen new rows are rows 16 (q mod 7) .. + 15 -- a block that never wraps inside, so a
    // lane's row address is one addition -- and a tile's first (carried) row lies timeRange - 1 rows in front of its first new one;
    // no copying between tiles
    int rbase = wrap(kTile - (T - 1));
    int wrow = 0;                                                    // the first ring row of the next sub-tile to finish columns of
    auto blk_of = [&](int64_t uu) { return fbase + 16 * uu + (R - 1); };   // first block of sub-tile uu (frames end on their last block)

    // prologue: sub-tiles 0, 1 and 2 loaded, the maxima of the first two taken, sub-tile 0 folded.  In the loop a sub-tile's
    // samples are loaded three iterations ahead of its MFMAs, its maxima taken two ahead, its fragments made one ahead: nothing
    // waits for memory inside an iteration.
    Raw r1 = load_raw(blk_of(0)), r2 = load_raw(blk_of(1));
    Raw r3 = load_raw(blk_of(2));
    __syncthreads();                                                 // (the zeroed maxima)
    raw_max(r1, bmax + 0);
    raw_max(r2, bmax + 16);
    __syncthreads();
    fold_store(r1, 0, bfr0, true);
    r1 = r2;                                                         // r1: sub-tile u + 1 (to fold), r2: sub-tile u + 2 (to take the maxima of)
    r2 = r3;

    // ---- the stages of an iteration.  They are independent of each other (each works on a different sub-tile), so the two
    // waves of a SIMD -- w and w + 4 -- take them in different orders: one multiplies while the other folds or finishes columns.
    // The window stage comes before the multiplication in both: it reads what the last multiplication left in registers.
    auto stage_load_fold = [&](bool fh) {                            // raw samples of sub-tile u + 3; fragments of sub-tile u + 1 (fh: a wave of the first half)
        if (fh && tid < 16) bmax[((u + 3) & 3) * 16 + tid] = 0u;           // (sub-tile u - 1's slot: read for the last time before this barrier)
#ifdef SYLDET_B_NOLOAD
        r3 = load_raw(-100000);
#else
        r3 = load_raw(blk_of(u + 3));                                // (past the recording: zeros from the descriptor's bounds check)
#endif
#ifndef SYLDET_B_NOFOLD
        fold_store(r1, (int)((u + 1) & 3), (u & 1) ? bfr0 : bfr1, fh);
#endif
    };
    // the first k-step's fragments are asked for before the window stage, whose arithmetic then covers their way out of LDS
    // (SYLDET_B_PREFETCH k-steps: 1 ships)
#ifndef SYLDET_B_PREFETCH
#define SYLDET_B_PREFETCH 1                       // (k-steps of fragments asked for ahead: 1 is -0.9 %, 2 is -0.7 %: MEASUREMENTS R5.4)
#endif
    uint32x4 pf[SYLDET_B_PREFETCH > 0 ? SYLDET_B_PREFETCH : 1][4] = {};
    auto mfma_prefetch = [&]() {
        const uint32x4 *bf = reinterpret_cast<const uint32x4 *>((u & 1) ? bfr1 : bfr0) + lane;
#pragma unroll
        for (int q = 0; q < SYLDET_B_PREFETCH; q++)
#pragma unroll
            for (int w = 0; w < 4; w++) pf[q][w] = bf[(w * KS + q) * 64];
    };
    auto stage_mfma = [&]() {                                        // sub-tile u: B', the sliding sum, the edge bins
        const int par = (int)(u & 1);
        // ---- B'_n[k] for this wave's bins: cosine rows against the sums, sine rows against the differences
        const uint32x4 *bf = reinterpret_cast<const uint32x4 *>(par ? bfr1 : bfr0) + lane;
        const floatx4 rec = recs[(u & 3) * 16 + n];
        floatx4 are = {0.f, 0.f, 0.f, 0.f}, aim = are;
        uint32x4 fr[2][4];                                           // fragments two k-steps ahead of their MFMAs
#pragma unroll
        for (int w = 0; w < 4; w++) fr[0][w] = SYLDET_B_PREFETCH >= 1 ? pf[0][w] : bf[(w * KS + 0) * 64];
        if (KS > 1) {
#pragma unroll
            for (int w = 0; w < 4; w++) fr[1][w] = SYLDET_B_PREFETCH >= 2 ? pf[SYLDET_B_PREFETCH >= 2 ? 1 : 0][w] : bf[(w * KS + 1) * 64];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const half8 bsh = as_half8(fr[ks & 1][0]), bsl = as_half8(fr[ks & 1][1]);
            const half8 bdh = as_half8(fr[ks & 1][2]), bdl = as_half8(fr[ks & 1][3]);
            are = mfma(ac[ks][0], bsh, are);
            aim = mfma(as_[ks][0], bdh, aim);
            are = mfma(ac[ks][0], bsl, are);
            aim = mfma(as_[ks][0], bdl, aim);
            are = mfma(ac[ks][1], bsh, are);
            aim = mfma(as_[ks][1], bdh, aim);
            if (ks + 2 < KS) {
#pragma unroll
                for (int w = 0; w < 4; w++) fr[ks & 1][w] = bf[(w * KS + ks + 2) * 64];
            }
        }
        // the block's first sample (its real part; the imaginary part rode in slot 0), then back to true units
        const float x0s = rec[0], un = rec[1];
        floatx4 bre, bim;
#pragma unroll
        for (int i = 0; i < 4; i++) {
#ifdef SYLDET_B_NOUNSCALE                            // (knock-out, wrong results: what the blocks' way back to true units costs -- the bound on folding it elsewhere)
            bre[i] = fmaf(cre[i], x0s, are[i]);
            bim[i] = aim[i];
            (void)un;
#else
            bre[i] = fmaf(cre[i], x0s, are[i]) * un;
            bim[i] = aim[i] * un;
#endif
#ifndef SYLDET_B_CONTRACT
            // (kept out of the compiler's sight as products: contracted into the sliding sum's additions they would take the
            // shifts' place in them -- a move and a multiply-add where one shifted addition does)
            asm volatile("" : "+v"(bre[i]), "+v"(bim[i]));
#endif
        }
        // ---- frames end on their last block.  Four blocks a frame: Y'_n = sum_{q' < 4} rho^(3 - q') B'_{n - q'} with rho = (-i)^k, k = i (mod 4):
        // 1, -i, -1, i for i = 0 .. 3.  In two levels: V_n = rho B'_n + B'_{n-1}, Y'_n = rho^2 V_n + V_{n-2} -- two shifts a value
        // instead of three (the previous sub-tile's last columns carry: B' for the first level, V for the second).
        floatx4 vre = bre, vim = bim;
        if (R == 4) {
#ifndef SYLDET_B_SUM_PAIRS
            // (in phases: the eight first halves, then the eight second halves -- a shifted addition that reads the register the
            // instruction before wrote waits two states; eight independent ones apart it waits none)
            const float l1[8] = {bre[0], bim[0], bim[1], -bre[1], -bre[2], -bim[2], -bim[3], bre[3]};
            const float c1[8] = {bre[0], bim[0], bre[1], bim[1], bre[2], bim[2], bre[3], bim[3]};
            const float p1[8] = {bre_prev[0], bim_prev[0], bre_prev[1], bim_prev[1], bre_prev[2], bim_prev[2], bre_prev[3], bim_prev[3]};
            float t1[8];
#pragma unroll
            for (int k = 0; k < 8; k++) t1[k] = l1[k] + shr_cur<1>(c1[k]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; k++) t1[k] = t1[k] + shl_prev<1>(p1[k]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; i++) { vre[i] = t1[2 * i]; vim[i] = t1[2 * i + 1]; }
            const float p2[8] = {vre_prev[0], vim_prev[0], vre_prev[1], vim_prev[1], vre_prev[2], vim_prev[2], vre_prev[3], vim_prev[3]};
            float t2[8];
#pragma unroll
            for (int k = 0; k < 8; k++) t2[k] = ((k >> 1) & 1 ? -t1[k] : t1[k]) + shr_cur<2>(t1[k]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; k++) t2[k] = t2[k] + shl_prev<2>(p2[k]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; i++) { yre[i] = t2[2 * i]; yim[i] = t2[2 * i + 1]; }
#else
            vre[0] = back_add<1>(bre[0], bre[0], bre_prev[0]);    vim[0] = back_add<1>(bim[0], bim[0], bim_prev[0]);     // rho = 1
            vre[1] = back_add<1>(bim[1], bre[1], bre_prev[1]);    vim[1] = back_add<1>(-bre[1], bim[1], bim_prev[1]);    // -i z = (b, -a)
            vre[2] = back_add<1>(-bre[2], bre[2], bre_prev[2]);   vim[2] = back_add<1>(-bim[2], bim[2], bim_prev[2]);    // -z
            vre[3] = back_add<1>(-bim[3], bre[3], bre_prev[3]);   vim[3] = back_add<1>(bre[3], bim[3], bim_prev[3]);     // i z = (-b, a)
            yre[0] = back_add<2>(vre[0], vre[0], vre_prev[0]);    yim[0] = back_add<2>(vim[0], vim[0], vim_prev[0]);     // rho^2 = 1
            yre[1] = back_add<2>(-vre[1], vre[1], vre_prev[1]);   yim[1] = back_add<2>(-vim[1], vim[1], vim_prev[1]);    // -1
            yre[2] = back_add<2>(vre[2], vre[2], vre_prev[2]);    yim[2] = back_add<2>(vim[2], vim[2], vim_prev[2]);     // 1
            yre[3] = back_add<2>(-vre[3], vre[3], vre_prev[3]);   yim[3] = back_add<2>(-vim[3], vim[3], vim_prev[3]);    // -1
#endif
        } else if (R == 2) {
            // two blocks a frame: Y'_n = rho B'_n + B'_{n-1} with rho = (-1)^k
#pragma unroll
            for (int i = 0; i < 4; i++) {
                yre[i] = back_add<1>((i & 1) ? -bre[i] : bre[i], bre[i], bre_prev[i]);
                yim[i] = back_add<1>((i & 1) ? -bim[i] : bim[i], bim[i], bim_prev[i]);
            }
        } else {
            yre = bre; yim = bim;                                    // (a frame is its block: not instantiated -- the FFT kernels are faster there)
        }
        upc = rec[2];
        dnc = rec[3];
        // this lane group's edge bins -> LDS for its neighbours
        edges[(par * 32 + 4 * wave + g) * 16 + n] = floatx4{yre[0], yim[0], yre[3], yim[3]};
        bre_prev = bre; bim_prev = bim; vre_prev = vre; vim_prev = vim;
    };
    floatx4 eLp = {0.f, 0.f, 0.f, 0.f}, eRp = eLp;
    auto window_prefetch = [&]() {                                   // the neighbours' edge bins asked for a stage ahead (-0.4 %)
        const int pe = (int)((u - 1) & 1);
        const int G = 4 * wave + g;
        eLp = G > 0 ? edges[(pe * 32 + G - 1) * 16 + n] : floatx4{0.f, 0.f, 0.f, 0.f};
        eRp = G < 31 ? edges[(pe * 32 + G + 1) * 16 + n] : floatx4{0.f, 0.f, 0.f, 0.f};
    };
    auto stage_window = [&](bool fh) {                                       // sub-tile u - 1 (ring rows wrow ..): window taps, |X|, columns
        const int pe = (int)((u - 1) & 1);                           // (u: this iteration's sub-tile, or one past the tile's last in the drain iteration)
        const int G = 4 * wave + g;
        // (a wave of the second half asked for them before its fold: window_prefetch)
        const floatx4 eL = fh ? (G > 0 ? edges[(pe * 32 + G - 1) * 16 + n] : floatx4{0.f, 0.f, 0.f, 0.f}) : eLp;
        const floatx4 eR = fh ? (G < 31 ? edges[(pe * 32 + G + 1) * 16 + n] : floatx4{0.f, 0.f, 0.f, 0.f}) : eRp;
        const float reL[4] = {eL[2], yre[0], yre[1], yre[2]}, imL[4] = {eL[3], yim[0], yim[1], yim[2]};
        const float reR[4] = {yre[1], yre[2], yre[3], eR[0]}, imR[4] = {yim[1], yim[2], yim[3], eR[1]};
        float cv[4], ssq = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float xr = fmaf(wA[i], yre[i], fmaf(wC[i], reL[i] + reR[i], -wS[i] * (imL[i] - imR[i])));
            const float xi = fmaf(wA[i], yim[i], fmaf(wS[i], reL[i] - reR[i], wC[i] * (imL[i] + imR[i])));
            cv[i] = __builtin_amdgcn_sqrtf(fmaf(xr, xr, xi * xi));        // zvabs / 2, :329-333 (Y is the DFT itself, not twice it)
            const bool inb = (unsigned)(fb0 + i) < (unsigned)F;
            if (scaling != 0) {
                // ln x / 20 log10 x of the band's columns (SyllableDetector.swift:185-207) through the hardware's base-2
                // logarithm (a denormal argument is lifted into its range first); ln 0 = -inf as in the reference.  Bins
                // outside the band meet zero weights: zeros, not logarithms of whatever is there
                const bool tiny = cv[i] < 0x1p-96f;
                const float l = __builtin_amdgcn_logf(tiny ? cv[i] * 0x1p64f : cv[i]) - (tiny ? 64.0f : 0.0f);
                cv[i] = inb ? l * lscale : 0.0f;
            }
#ifdef SYLDET_B_TAPS_UNIFORM
            ssq = inb ? fmaf(cv[i], cv[i], ssq) : ssq;
#else
            ssq = (!SC || inb) ? fmaf(cv[i], cv[i], ssq) : ssq;
#endif
        }
        ssq = xor32_sum(xor16_sum(ssq));
        const int rw = wrow + n;
        unsigned h0, l0, h1, l1;
        const float upw = scaling != 0 ? 16.0f : upc;               // (logarithms are within +-800: a fixed scale keeps them under f16's 65504)
#ifdef SYLDET_B_COLS_HI                              // (knock-out, 1e-3 results: the columns as ONE f16 each -- the bound on cheaper column formats)
        h0 = cvt_pk(cv[0] * upw, cv[1] * upw); l0 = 0u;
        h1 = cvt_pk(cv[2] * upw, cv[3] * upw); l1 = 0u;
#else
        split2(cv[0] * upw, cv[1] * upw, h0, l0);
        split2(cv[2] * upw, cv[3] * upw, h1, l1);
#endif
        // column index = bin - kb0 (the first layer's fragments are in that order); the lane's share of the address is loop-invariant
        *reinterpret_cast<uint32x2 *>(colh + wrow * CS + lane_col) = uint32x2{h0, h1};
#ifndef SYLDET_B_COLS_HI
        *reinterpret_cast<uint32x2 *>(coll + wrow * CS + lane_col) = uint32x2{l0, l1};
#endif
        if (g == 0) ssf8[wrow * kWaves + lane_ssf] = ssq;
        if (fh && wave == 0 && g == 0) fsc[rw] = scaling != 0 ? 0.0625f : dnc;
    };

#ifdef SYLDET_B_STAMPS
    unsigned long long tsum[8] = {0}, tk = 0;
#define SD_BT(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now = __builtin_amdgcn_s_memtime(); tsum[i] += now - tk; tk = now; __builtin_amdgcn_sched_barrier(0); }
    tk = __builtin_amdgcn_s_memtime();
#else
#define SD_BT(i)
#endif
    // (SYLDET_B_NO*: diagnostic builds with one stage knocked out, tools/knockouts.sh; never the shipped library)
#if defined(SYLDET_B_NOMFMA) || defined(SYLDET_B_NOMW)
    constexpr bool kDoM = false;
#else
    constexpr bool kDoM = true;
#endif
#if defined(SYLDET_B_NOWINDOW) || defined(SYLDET_B_NOMW)
    constexpr bool kDoW = false;
#else
    constexpr bool kDoW = true;
#endif
    const bool first_half = wave < kWaves / 2;
    // ---- a tile's end, in two stages that run inside the NEXT tile's second and third iterations (round 5; the last tile's after the
    // loop), so that the stream of sub-tiles never stops: every iteration multiplies one sub-tile and finishes the columns of the one
    // before, across tile boundaries, and a tile costs six barriers instead of nine.
    // (1) the frames' sums of squares from the waves' partial sums, in a fixed order, and the tap products of the tile's NEW rows,
    // P[(t, h), j] for all taps at once (three row tiles), back to true units; the carried rows' sums and products are the tile
    // before's, still in place (zeros in front of the first tile) -- and must be: the iteration that runs this stage is already
    // writing the next tile's first columns over the carried rows (the ring has 112 rows, a tile uses 107).
    // (2) the evaluations, an iteration later: they read sums and products only.
    int rbase_end = 0, tr_end = -1;                                  // the ending tile's first ring row and index (none yet)
    auto stage_taps = [&]() {
        auto is_new = [&](int row) {                                 // a ring row the ending tile has made itself
            const int rel = row - rbase_end + (row < rbase_end ? kTile : 0);
            return (unsigned)(rel - (T - 1)) < (unsigned)kNew;
        };
        if (tid < kTile && is_new(tid)) {
            float a = 0.0f;
#pragma unroll
            for (int w = 0; w < kWaves; w++) a += ssf8[tid * kWaves + w];
            ssf[tid] = a;
        }
#ifndef SYLDET_B_NOTAPS
        if (wave < kTile / 16) {
            const int fr = 16 * wave + n;
            const _Float16 *bph = colh + fr * CS + 8 * g, *bpl = coll + fr * CS + 8 * g;
            floatx4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kb = 0; kb < KB; kb++) {
                const half8 bh = as_half8(*reinterpret_cast<const uint32x4 *>(bph + 32 * kb));
                const half8 bl = as_half8(*reinterpret_cast<const uint32x4 *>(bpl + 32 * kb));
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    const half8 ah = as_half8(afr[((m * KB + kb) * 2 + 0) * 64 + lane]), al = as_half8(afr[((m * KB + kb) * 2 + 1) * 64 + lane]);
                    acc[m] = mfma(ah, bh, acc[m]);
#ifndef SYLDET_B_COLS_HI
                    acc[m] = mfma(ah, bl, acc[m]);
                    acc[m] = mfma(al, bh, acc[m]);
#endif
                }
            }
            const float dn = fsc[fr];
            if (is_new(fr)) {                                        // (the other rows of the tile: half-written columns, nobody's products)
#pragma unroll
                for (int m = 0; m < 3; m++) *reinterpret_cast<floatx4 *>(pbuf + fr * PS + 4 * (4 * m + g)) = acc[m] * dn;
            }
        }
#endif
    };
    auto stage_evaluate = [&]() {
#ifndef SYLDET_B_NOEVAL
        // four threads an evaluation: row r starts the window of evaluation fbase + kNew tr + r - (T - 1); thread j of the four
        // takes taps j, j + 4, j + 8 and then hidden unit j (quad permutes carry the sums)
        // (tried: on the first-half waves alone, which reach the barrier with time to spare -- the waits even out, 685 / 198 -> 620 / 665
        // clocks an iteration, and the kernel is 0.5 % slower: MEASUREMENTS R5.4)
        if (tid < 4 * kNew) {
            const int r = (tid >> 6) * 16 + (int)((kEvalOrder >> (((tid >> 2) & 15) * 4)) & 15ull), j = tid & 3;
            floatx4 z = {0.f, 0.f, 0.f, 0.f};
            float ssw = 0.0f;
#pragma unroll
            for (int tt = 0; tt < 3; tt++) {
                const int t = j + 4 * tt;
                if (t < T) {
                    const int rr = wrap(rbase_end + r + t);
                    z += *reinterpret_cast<const floatx4 *>(pbuf + rr * PS + 4 * t);
                    ssw += ssf[rr];
                }
            }
            auto quad_sum = [](float v) {
                v += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
                v += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
                return v;
            };
            ssw = quad_sum(ssw);
            float zs[4];
#pragma unroll
            for (int h = 0; h < 4; h++) zs[h] = quad_sum(z[h]);
            const float alpha = d.w_unscale * __builtin_amdgcn_rsqf(ssw);
            // TanSig hidden unit j (rows past H meet zero weights), linear output
            const float zu = j == 0 ? zs[0] : (j == 1 ? zs[1] : (j == 2 ? zs[2] : zs[3]));
            const float a = fmaf(alpha, zu, b0[j == 0 ? 0 : (j == 1 ? 1 : (j == 2 ? 2 : 3))]);
            const float th = fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a * 2.885390081777927f) + 1.0f), 1.0f);
            // the second layer in the reference's order of units: y = b1 + w1[0] th0 + w1[1] th1 + ... (every lane of the four forms it alike)
            float thq[4];
#pragma unroll
            for (int h = 0; h < 4; h++) thq[h] = quad_sum(j == h ? th : 0.0f);
            float y = d.b1;
#pragma unroll
            for (int h = 0; h < 4; h++) y = fmaf(w1[h], thq[h], y);
            y = (y - d.oa) / d.og + d.ob;
            const int64_t ev = fbase + (int64_t)kNew * tr_end + r - (T - 1);
            const bool st = j == 0 && ev >= E0 && ev < E1;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), out_rs, st ? (unsigned)ev * 4u : 0xFFFFFFFFu, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)((double)y >= thr ? 1 : 0), flg_rs, st ? (unsigned)ev : 0xFFFFFFFFu, 0, 0);
        }
#endif
    };

    // One iteration: a barrier, then this wave's stages.  MM: there is a sub-tile to multiply (and to load and fold for);
    // WW: there is one to finish columns of -- all but the run's first iteration; the run's last has nothing else.  END: which stage
    // of the tile before's end this iteration carries (1: sums and tap products; 2: evaluations).  Straight-line bodies, no
    // per-stage conditions in the loop.
    // bfr[u & 1] holds sub-tile u's fragments, bmax[(u + 1) & 3] sub-tile u + 1's maxima, edges[(u - 1) & 1] sub-tile u - 1's edge bins
    auto iteration = [&](auto mm_, auto ww_, auto end_) {
        constexpr bool MM = decltype(mm_)::value, WW = decltype(ww_)::value;
        constexpr int END = decltype(end_)::value;
        SD_BT(0)
        __syncthreads();
        SD_BT(1)
        if (END == 2 && tr_end >= 0) stage_evaluate();
        SD_BT(5)
        if (first_half) {
            if (MM) mfma_prefetch();
            if (WW && kDoW) stage_window(true);
            SD_BT(4)
            if (END == 1 && tr_end >= 0) stage_taps();
            SD_BT(5)
            if (MM && kDoM) stage_mfma();
            SD_BT(2)
            if (MM) stage_load_fold(true);
            SD_BT(3)
        } else {
#ifdef SYLDET_B_ORDER_WLM                            // (diagnostic: window, fold, multiply in the second half too)
            if (WW && kDoW) window_prefetch();
            if (WW && kDoW) stage_window(false);
            SD_BT(4)
            if (END == 1 && tr_end >= 0) stage_taps();
            SD_BT(5)
            if (MM) mfma_prefetch();
            if (MM) stage_load_fold(false);
            SD_BT(3)
            if (MM && kDoM) stage_mfma();
            SD_BT(2)
#else
            if (WW && kDoW) window_prefetch();
            if (MM) stage_load_fold(false);
            SD_BT(3)
            if (MM) mfma_prefetch();
            if (WW && kDoW) stage_window(false);
            SD_BT(4)
            if (END == 1 && tr_end >= 0) stage_taps();
            SD_BT(5)
            if (MM && kDoM) stage_mfma();
            SD_BT(2)
#endif
        }
        if (WW) wrow = wrap(wrow + 16);
        if (MM) {
            // ---- sub-tile u + 2's block maxima (its samples were loaded an iteration ago); the stream moves on
            raw_max(r2, bmax + ((u + 2) & 3) * 16);
            r1 = r2;
            r2 = r3;
            u++;
        }
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    using end0 = std::integral_constant<int, 0>;
    using end1 = std::integral_constant<int, 1>;
    using end2 = std::integral_constant<int, 2>;
#ifndef SYLDET_B_NO_PREWAIT
    __builtin_amdgcn_s_waitcnt(0x0F70);              // (the compiler is told that the prologue's loads are complete: kernels_fused_s.hip has the story)
#endif
    // (round 6 experiments, MEASUREMENTS R6.2: a static priority for one half of the workgroup -- MI355X_MICROARCH.md, "Two waves per
    // SIMD", item 4: the second-dispatched half loses every arbitration)
#if defined(SYLDET_B_SETPRIO_SECOND)
    if (!first_half) __builtin_amdgcn_s_setprio(1);
#elif defined(SYLDET_B_SETPRIO_FIRST)
    if (first_half) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef SYLDET_B_END_INLINE                           // (diagnostic: the tile's end between the tiles, as through round 4)
    for (int tr = 0; tr < tiles; tr++) {
        iteration(yes{}, no{}, end0{});
        for (int s = 1; s < kSubs; s++) iteration(yes{}, yes{}, end0{});
        iteration(no{}, yes{}, end0{});
        rbase_end = rbase; tr_end = tr;
        __syncthreads();
        stage_taps();
        __syncthreads();
        stage_evaluate();
        rbase = wrap(rbase + kNew);
    }
#else
    if (tiles > 0) iteration(yes{}, no{}, end0{});   // the run's first sub-tile: nothing to finish columns of yet
    for (int tr = 0; tr < tiles; tr++) {
        if (tr > 0) iteration(yes{}, yes{}, end0{});
        iteration(yes{}, yes{}, end1{});
        iteration(yes{}, yes{}, end2{});
#ifndef SYLDET_B_NOUNROLL                            // (unrolled: the three sets of raw samples in flight rotate by renaming, not by copies: -1.3 %)
#pragma unroll
#endif
        for (int s = 3; s < kSubs; s++) iteration(yes{}, yes{}, end0{});
        rbase_end = rbase; tr_end = tr;
        rbase = wrap(rbase + kNew);                    // the last T - 1 rows are the next tile's first: the ring moves on, nothing is copied
    }
    if (tr_end >= 0) {                               // the last sub-tile's columns, the last tile's end
        iteration(no{}, yes{}, end0{});
        SD_BT(0)
        __syncthreads();
        SD_BT(1)
        stage_taps();
        __syncthreads();
        stage_evaluate();
        SD_BT(5)
    }
#endif
#ifdef SYLDET_B_STAMPS
    if (lane == 0 && (wave == 0 || wave == 4))
        for (int i = 0; i < 8; i++) atomicAdd(&g_bdft_stamps[(wave ? 8 : 0) + i], tsum[i]);
#endif
#undef SD_BT
}

}  // namespace

hipError_t launch_bdft_net(const MlpxDesc &d, const BdftDesc &bd, const float *samples, int64_t stride, int C, int64_t S, int64_t J, int64_t E,
                           float *outputs, uint8_t *flags, hipStream_t stream)
{
    if (E <= 0 || C <= 0) return hipSuccess;
    if ((uint64_t)E * 4u >= 0xFFFFFFF0ull || (uint64_t)S * 4u >= 0x7fffffffull || (bd.R != 4 && bd.R != 2)) return hipErrorInvalidValue;
    // a workgroup walks a contiguous run of one channel; runs as long as still leaves two rounds of workgroups on the 256 CUs
    int64_t runs_per_channel = (512 + C - 1) / C;
    const int64_t min_run = 4 * kNew;
    runs_per_channel = std::max<int64_t>(1, std::min<int64_t>(runs_per_channel, (E + min_run - 1) / min_run));
    const int64_t evals_per_run = (E + runs_per_channel - 1) / runs_per_channel;
    const int64_t runs = (E + evals_per_run - 1) / evals_per_run;
    dim3 grid((unsigned)runs, (unsigned)C);
    const int KS = bd.hop / 64;
    const int frag = 4 * KS * 1024 < 16384 ? 16384 : 4 * KS * 1024;
    const int lds = 3 * 4 * 2 * 1024 + 2 * kTile * d.col_stride * 2 + 2 * kTile * 4 + 2 * frag + 2 * 16 * 32 * 16 + kTile * kWaves * 4 + 4 * 16 * 4 + 4 * 16 * 16 +
                    kTile * kPS * 4;
    if (lds > 160 * 1024 || d.T > 12 || d.T - 1 + kNew > kTile) return hipErrorInvalidValue;
#ifdef SYLDET_B_STAMPS
#define SD_BDFT_STAMP_REPORT                                                                                                   \
    {                                                                                                                          \
        unsigned long long hs[16];                                                                                             \
        hipStreamSynchronize(stream);                                                                                          \
        hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_bdft_stamps), sizeof(hs));                                                        \
        const double wg = (double)runs * C;                                                                                    \
        static const char *nm[8] = {"work before the barrier -> arrival", "barrier wait", "MFMA + sliding sum + edges", "load + fold", "window + columns", "tile end (sums, taps, evaluation, carry)", "-", "-"}; \
        for (int w = 0; w < 2; w++)                                                                                            \
            for (int i = 0; i < 6; i++) std::fprintf(stderr, "[bdft stamps] wave %d  %-42s %10.0f cycles per workgroup\n", 4 * w, nm[i], (double)hs[8 * w + i] / wg); \
        unsigned long long z[16] = {0};                                                                                        \
        hipMemcpyToSymbol(HIP_SYMBOL(g_bdft_stamps), z, sizeof(z));                                                            \
    }
#else
#define SD_BDFT_STAMP_REPORT
#endif
#define SD_BDFT_GO(KS_)                                                                                                        \
    if (KS == KS_) {                                                                                                           \
        auto kern = bd.R == 4 ? (d.scaling != 0 ? bdft_net_kernel<KS_, true, 4> : bdft_net_kernel<KS_, false, 4>)                      \
                              : (d.scaling != 0 ? bdft_net_kernel<KS_, true, 2> : bdft_net_kernel<KS_, false, 2>);                     \
        hipError_t st = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);              \
        if (st != hipSuccess) return st;                                                                                       \
        hipLaunchKernelGGL(kern, grid, dim3(kBlock), (size_t)lds, stream, d, bd, samples, stride, S, J, E, evals_per_run, outputs, flags); \
        SD_BDFT_STAMP_REPORT                                                                                                   \
        return hipGetLastError();                                                                                              \
    }
    SD_BDFT_GO(4) SD_BDFT_GO(2)
#undef SD_BDFT_GO
    return hipErrorInvalidValue;
}

}  // namespace sd
