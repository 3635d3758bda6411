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
// MI355X formulation (not a translation of the vDSP call sequence): two chained GEMMs on the matrix
// cores (v_mfma_f32_16x16x32_f16) whose B operands are *addresses*, not copies.
//   1. The detector needs F (<= 32) bins of each N-point spectrum, so the windowed DFT of a tile of
//      16 frames is  Xt[64 rows, 16 frames] = Dt[rows, W] . S[W, 16 frames]  with
//      Dt = window o {cos, -sin} (32 real + 32 imaginary rows); column j of S is simply the staged
//      sample stream at offset j*hop (no per-frame copy, no ring).  Every operand is split into f16
//      hi + lo (block floating point, power-of-two scales): hi*hi + hi*lo + lo*hi reproduces an fp32
//      product to ~2^-21 and the fp32 accumulate keeps the sum; measured error against the fp64
//      anchor is below an fp32 FFT's.  The samples are split once, when they are staged
//      (v_fma_mixlo/mixhi_f16: two VALU instructions per sample, the block scale rides in the FMA).
//   2. |X| columns (f16 hi/lo) go to a small LDS buffer [frame][bin].  The first network layer,
//      folded with the affine input maps into W' = W0 o gain, is  Z[h, e] = sum_t W'_t[h, :] . C[:, e+t]:
//      again a GEMM (K = 32 bins per tap) whose B operand for tap t is the column buffer at row
//      offset e + t.  The normalisers' per-frame statistics (sum of squares, min/max, mean/M2) sit in a
//      small fp32 array next to the columns and are combined per window.  An evaluation then finishes
//      in registers: scale, transfer function, second layer, reverse map, threshold.
// Every frame is transformed once (the reference re-reads each column T times).
//
// Workgroup = 8 waves x 16 frames = one 128-frame pass at a time over the workgroup's segment of one
// channel, all waves in the same phase (see fused_kernel).  LDS -- 64 KB of basis fragments + 68 KB of
// staged samples + 22 KB of columns -- allows one workgroup per CU.  HBM traffic = every sample once
// (+ W - hop samples of overlap per pass, L2 hits) + 5 bytes per evaluation.  Buffer loads with hardware
// bounds instead of guards, issued a whole pass ahead of their use.
//
// gfx950 only.  wave = 64.

#include <cstdlib>

#include "fused_common.hpp"

namespace sd {

namespace {

using namespace fused_dev;

constexpr int kBlock = kFusedBlock;            // 512 threads = 8 waves
constexpr int kWaves = kBlock / 64;
constexpr int kPass = kFusedTileFrames;        // 128 frames per pass = 16 per wave
constexpr int kColStride = kFusedColStride;

// Diagnostic variant (-DSYLDET_ASM_PREFETCH; not the shipped build -- measured 4 % slower):
// the fragment fetches of one k-step as volatile instructions, issued where they are written (a whole k-step ahead
// of their MFMAs), not where the scheduler would sink them.  The data is in flight until frags_wait.
template <int KSN>
[[maybe_unused]] __device__ __forceinline__ void frags_fetch(unsigned a_addr, unsigned bh_addr, unsigned bl_addr, uint32x4 (&a)[8], uint32x4 &bh, uint32x4 &bl)
{
    asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(bh) : "v"(bh_addr));
    asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(bl) : "v"(bl_addr));
#pragma unroll
    for (int i = 0; i < 8; i++)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[i]) : "v"(a_addr), "n"((KSN * 8 + i) * 1024));
}
[[maybe_unused]] __device__ __forceinline__ void frags_wait(uint32x4 (&a)[8], uint32x4 &bh, uint32x4 &bl)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(bh), "+v"(bl));
}

// Diagnostic stamps (STAMP instantiation only; never the shipped path): s_memtime at phase boundaries,
// read back (one wait) at the end of the pass so the stamps do not drain the memory pipelines in between.
#define SD_TICK(slot)                                                                      \
    if (STAMP) {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        tick[slot] = __builtin_amdgcn_s_memtime();                                         \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    }

// KS: k-steps of 32 samples; TMAX / NL: array sizes for taps and staging quads; EXACT: timeRange == TMAX
// and nload == NL are compile-time facts (no guards); SKEW: staged samples carry bank-spreading padding;
// LEAN: the configuration class of the reference's example detector is a compile-time fact --
// l2normalize first, linear |X| columns, two layers, TanSig hidden units (at most 4), one output --
// so that instantiation carries only the code it runs.  KNOCK: diagnostic knock-out mask (0 in every
// shipped instantiation).
//
// One pass = 128 frames = 16 per wave, all eight waves in the same phase (two per SIMD: while one waits for
// LDS the other feeds the matrix pipe -- a single wave cannot, its LDS fetch rate is capped at ~32 B/clk):
//     block M:  DFT of the wave's 16 frames from the staged samples (f16 hi + lo, block floating point)
//               || evaluation of the PREVIOUS pass (first layer as a shifted GEMM over the column buffer, rest of
//                  the network in registers, stores), cut into steps that sit between the k-steps -- branch-free,
//                  so the whole block is one scheduling region
//               || block-max partial of the NEXT pass's samples (already in registers)
//   barrier
//     previous pass's last T-1 columns -> transition strip;  magnitudes of this pass -> column buffer (own scale,
//     first T-1 also into the strip);
//     next pass's samples: scale, split, -> LDS (the staged region is free now);  issue the loads of the pass
//     after that (a whole pass of lead time)
//   barrier
template <int KS, int TMAX, int NL, bool EXACT, bool SKEW, bool LEAN, bool STAMP, int KNOCK, bool SPECT>
__global__ void __launch_bounds__(kBlock, 2)
fused_kernel(const FusedDesc d, const float *__restrict__ samples, int64_t stride, int64_t s_eff, int64_t E,
             float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32x4 *lds_dfrag = reinterpret_cast<uint32x4 *>(smem + d.lds_dfrag);
    float *red = reinterpret_cast<float *>(smem + d.lds_red);        // [8 waves] block-max partials
    float *cst = reinterpret_cast<float *>(smem + d.lds_cst);
    constexpr int kom = KNOCK;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = lane & 15;          // frame (DFT) / evaluation (first layer) column inside the wave's tile
    const int g4 = lane >> 4;         // k block 8*g4..8*g4+7 of an operand; rows 4*g4..4*g4+3 of a result
    const int c = blockIdx.y;
    const int64_t e_b = (int64_t)blockIdx.x * d.seg_evals;
    if (e_b >= E) return;
    const int64_t e_e = (e_b + d.seg_evals < E) ? e_b + d.seg_evals : E;
    const float *row = samples + (int64_t)c * stride;
    const int PS = d.ps, T = EXACT ? TMAX : d.T, H = d.H;      // PS: column slots = 2 (T - 1) transition slots + 128
    const int XS = 2 * (T - 1);                                 // the pass's own columns start at slot XS
    const int nload = EXACT ? NL : d.nload;
    const int norm = LEAN ? 1 : d.norm, scaling = LEAN ? 0 : d.scaling;
    const int n_layers = LEAN ? 2 : d.n_layers, n_out = LEAN ? 1 : d.n_out, tf0 = LEAN ? 0 : d.tf0, tf1 = LEAN ? 2 : d.tf1;
    const int fl = 16 * wave + f;     // this lane's frame / evaluation slot inside the pass
    const int runs = d.runs;

    _Float16 *smph = reinterpret_cast<_Float16 *>(smem + d.lds_smp);    // staged samples of one pass, f16 hi
    _Float16 *smpl = smph + d.smp_stride;                                //                             f16 lo
    _Float16 *colh = reinterpret_cast<_Float16 *>(smem + d.lds_colh);   // [PS][kColStride] |X| columns, hi parts
    _Float16 *coll = reinterpret_cast<_Float16 *>(smem + d.lds_coll);   //                               lo parts
    float *stat = reinterpret_cast<float *>(smem + d.lds_stat);          // [2][PS] per-frame min/max or mean/M2

    // ---- once per workgroup: constants
    for (int i = tid; i < KS * 8 * 64; i += kBlock) lds_dfrag[i] = reinterpret_cast<const uint32x4 *>(d.dfrag)[i];
    if (!SPECT && tid < 16) reinterpret_cast<double *>(cst + kCstThr)[tid] = tid < n_out ? d.thresholds[tid] : 0.0;
    for (int i = tid; i < (SPECT ? 0 : d.n_out_fns * (1 + 2 * n_out)); i += kBlock) cst[kCstOut + i] = d.out_params[i];
    // first-layer fragments, one (hi, lo) pair per tap: A operand, lane l holds row l&15 (hidden unit, or the
    // statistic row), k = 8*(l>>4) + j (bin)
    half8 afr[TMAX][2];
#pragma unroll
    for (int t = 0; t < TMAX; t++)
#pragma unroll
        for (int p = 0; p < 2; p++)
            afr[t][p] = SPECT ? half8{} : as_half8(reinterpret_cast<const uint32x4 *>(d.afrag)[((t < T ? t : 0) * 2 + p) * 64 + lane]);
    // evaluation-phase constants of the 4 hidden units this lane group owns (rows 4*g4 + j of a result)
    float c_b0[4], c_rv[4], c_w1[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int h = 4 * g4 + j;
        c_b0[j] = (!SPECT && h < H) ? d.bias0[h] : 0.0f;
        c_rv[j] = (!SPECT && h < H) ? d.rvec[h] : 0.0f;
#pragma unroll
        for (int o = 0; o < 4; o++) c_w1[o][j] = (!SPECT && n_layers == 2 && h < H && o < n_out) ? d.w1[o * H + h] : 0.0f;
    }
    float c_b1[4];
#pragma unroll
    for (int o = 0; o < 4; o++) c_b1[o] = (!SPECT && n_layers == 2 && o < n_out) ? d.b1[o] : 0.0f;

    // results leave through bounds-checked descriptors of this channel's rows: a lane with nothing to store uses an
    // offset past the end (dropped by the hardware), so the evaluation carries no branches
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(
        outputs ? outputs + (int64_t)c * E * n_out : nullptr, 0, outputs ? (int)(E * n_out * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t flg_rs = __builtin_amdgcn_make_buffer_rsrc(flags ? flags + (int64_t)c * E : nullptr, 0, flags ? (int)E : 0, 0x00020000);
    // SPECT: the columns themselves leave, [C][J][F] fp32 (E counts frames then)
    const __amdgpu_buffer_rsrc_t spc_rs = __builtin_amdgcn_make_buffer_rsrc(
        SPECT ? d.spect_out + (int64_t)c * E * d.F : nullptr, 0, SPECT ? (int)(E * d.F * 4) : 0, 0x00020000);
    // LEAN: at most one output map, applied unconditionally (identity when there is none)
    float lean_oa = 0.0f, lean_og = 1.0f, lean_ob = 0.0f;
    if (LEAN && d.n_out_fns == 1) { lean_oa = d.out_params[0]; lean_og = d.out_params[1]; lean_ob = d.out_params[2]; }

    // this lane's frame in the staged stream, and where k-step ks of lane group g4 starts inside it
    // A k-step's 32 samples are four 8-sample blocks KS blocks apart (block ks + KS g4 for lane group g4), not four
    // neighbours: the lane groups of one ds_read then sit 4 KS dwords apart and, at hop 132, on disjoint banks.
    const int foff = fl * (d.hop + (SKEW ? d.skew : 0)) + (SKEW ? 0 : 8 * KS * g4);
    const _Float16 *fph = smph + foff, *fpl = smpl + foff;
    int ko[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) ko[ks] = SKEW ? d.koff[ks * 4 + g4] : 8 * ks;   // immediates without skew

    // raw samples of one pass: quads 4*(tid + 512 k), k < nload, through a bounds-checked descriptor
    uint32x4 v[NL];
    auto pass_rsrc = [&](int p) {
        return tile_rsrc(row, (e_b + (int64_t)kPass * p) * d.hop + d.gap, p < runs ? s_eff : 0, d.nsmp);
    };
    auto load_pass = [&](int p) {
        const __amdgpu_buffer_rsrc_t rs = pass_rsrc(p);
#pragma unroll
        for (int k = 0; k < NL; k++)
            if (k < nload && !(kom & 4)) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * tid + 16 * kBlock * k, 0, 0);
    };
    // block-max partial of the quads in v[] -> red
    auto max_partial = [&]() {
        float amax = 0.0f;
#pragma unroll
        for (int k = 0; k < NL; k++)
            if (k < nload) {
                const floatx4 q = as_floatx4(v[k]);
                amax = absmax3(absmax3(amax, q[0], q[1]), q[2], q[3]);
            }
        amax = wave_max_nonneg(amax);                        // v_max3 drops NaNs: amax is a plain non-negative number
        if (lane == 0) red[wave] = amax;
    };
    // block floating point: the pass's largest sample goes to [2^13, 2^14); returns the scale's exponent
    auto pass_scale = [&](floatx4 r0, floatx4 r1) {
        const float amax = fmaxf(fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[2], r0[3])), fmaxf(fmaxf(r1[0], r1[1]), fmaxf(r1[2], r1[3])));
        int e = 13 - (int)((__float_as_uint(amax) >> 23) & 0xffu) + 127;
        e = amax > 0.0f ? (e < -100 ? -100 : (e > 100 ? 100 : e)) : 0;
        return __builtin_amdgcn_readfirstlane(e);
    };
    // quads in v[] -> scaled, split into f16 hi + lo, -> the staged region
    auto stage_pass = [&](int e) {
        const float sx = pow2f(e);
#pragma unroll
        for (int k = 0; k < NL; k++)
            if (k < nload && !(kom & 2)) {
                const floatx4 q = as_floatx4(v[k]);
                const int i = 4 * (tid + kBlock * k);
                if (k + 1 < nload || i < d.nsmp) {             // only the last quad set can reach past the pass
                    unsigned h0, l0, h1, l1;
                    split_pair_scaled(q[0], q[1], sx, h0, l0);
                    split_pair_scaled(q[2], q[3], sx, h1, l1);
                    const int p = i + (SKEW ? d.skew * (int)__umulhi((unsigned)i, d.hop_magic) : 0);
                    uint32x2 uh = {h0, h1}, ul = {l0, l1};
                    *reinterpret_cast<uint32x2 *>(smph + p) = uh;
                    *reinterpret_cast<uint32x2 *>(smpl + p) = ul;
                }
            }
    };
    int se_next;                      // scale exponent of the pass staged most recently
    load_pass(0);
    max_partial();
    __syncthreads();
    se_next = pass_scale(*reinterpret_cast<const floatx4 *>(red), *reinterpret_cast<const floatx4 *>(red + 4));
    stage_pass(se_next);
    load_pass(1);                     // arrives during the first pass's matrix work
    int se = 0, cse = 0, se_prev = 0;   // sample / column scale exponents of this pass, sample scale of the one before
    unsigned long long tsum[16] = {0}, tick[8] = {0};
    if (STAMP) tick[5] = __builtin_amdgcn_s_memtime();
    __syncthreads();

#include "fused_eval.inc"
    int cse_post = 0, csx_post = 0;                           // column scales (own, transition strip) of the pass being evaluated
    for (int p = 0; p < runs; p++) {
        // ================= block M: DFT of pass p  ||  evaluation of pass p-1  ||  block max of pass p+1
        // band-limited DFT of this wave's 16 frames on the matrix cores: four 16-row tiles (re bins 0-15, re 16-31,
        // im 0-15, im 16-31), 12 MFMAs per k-step of 32 samples; the fragments of k-step ks+1 are fetched while the
        // MFMAs of k-step ks execute.  The evaluation of the previous pass (its columns are complete, pass -1 is a dry
        // run whose stores are masked) is cut into steps that fill the VALU / LDS slots between the MFMAs.
        floatx4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        {
#ifdef SYLDET_ASM_PREFETCH
            uint32x4 a[8], bhu, blu;
            const unsigned a_addr = lds_addr(lds_dfrag + lane);
            frags_fetch<0>(a_addr, lds_addr(fph + ko[0]), lds_addr(fpl + ko[0]), a, bhu, blu);
#pragma unroll
            for (int ks = 0; ks < ((kom & 256) ? 0 : KS); ks++) {
                frags_wait(a, bhu, blu);
                half8 ah[4], al[4];
#pragma unroll
                for (int m = 0; m < 4; m++) { ah[m] = as_half8(a[2 * m]); al[m] = as_half8(a[2 * m + 1]); }
                const half8 cbh = as_half8(bhu), cbl = as_half8(blu);
                if (ks + 1 < KS) {                                // fragments of the next k-step
                    const unsigned ha = lds_addr(fph + ko[ks + 1 < KS ? ks + 1 : 0]), la = lds_addr(fpl + ko[ks + 1 < KS ? ks + 1 : 0]);
                    switch (ks + 1) {
                    case 1: frags_fetch<1>(a_addr, ha, la, a, bhu, blu); break;
                    case 2: frags_fetch<2>(a_addr, ha, la, a, bhu, blu); break;
                    case 3: frags_fetch<3>(a_addr, ha, la, a, bhu, blu); break;
                    case 4: frags_fetch<4>(a_addr, ha, la, a, bhu, blu); break;
                    case 5: frags_fetch<5>(a_addr, ha, la, a, bhu, blu); break;
                    case 6: frags_fetch<6>(a_addr, ha, la, a, bhu, blu); break;
                    default: frags_fetch<7>(a_addr, ha, la, a, bhu, blu); break;
                    }
                }
#else
            half8 bh = lds_half8(fph + ko[0]), bl = lds_half8(fpl + ko[0]);
            uint32x4 a[8];
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = lds_dfrag[i * 64 + lane];
#pragma unroll
            for (int ks = 0; ks < ((kom & 256) ? 0 : KS); ks++) {
                half8 ah[4], al[4];
#pragma unroll
                for (int m = 0; m < 4; m++) { ah[m] = as_half8(a[2 * m]); al[m] = as_half8(a[2 * m + 1]); }
                const half8 cbh = bh, cbl = bl;
                if (ks + 1 < KS) {                                // fragments of the next k-step
#pragma unroll
                    for (int i = 0; i < 8; i++) a[i] = lds_dfrag[((ks + 1) * 8 + i) * 64 + lane];
                    bh = lds_half8(fph + ko[ks + 1]);
                    bl = lds_half8(fpl + ko[ks + 1]);
                }
#endif
                if (!(kom & 1)) {
#pragma unroll
                    for (int m = 0; m < 4; m++) acc[m] = mfma(ah[m], cbh, acc[m]);
#pragma unroll
                    for (int m = 0; m < 4; m++) acc[m] = mfma(ah[m], cbl, acc[m]);
#pragma unroll
                    for (int m = 0; m < 4; m++) acc[m] = mfma(al[m], cbh, acc[m]);
                } else {
#pragma unroll
                    for (int m = 0; m < 4; m++) asm volatile("" ::"v"(ah[m]), "v"(al[m]));
                    asm volatile("" ::"v"(cbh), "v"(cbl));
                }
                // the previous pass's evaluation, one or two steps per k-step
                if (!(kom & 128) && !SPECT) {
                    if (KS == 8) post_step(ks, p - 1, cse_post, csx_post);
                    else { post_step(2 * ks, p - 1, cse_post, csx_post); post_step(2 * ks + 1, p - 1, cse_post, csx_post); }
                }
            }
        }
        // block-max partial of the next pass's samples (fetched a pass ago)
        if (p + 1 < runs) max_partial();
        SD_TICK(0)
        __syncthreads();          // all reads of the staged samples and of the columns are done; partial maxima are in
        SD_TICK(1)
        const floatx4 red0 = *reinterpret_cast<const floatx4 *>(red), red1 = *reinterpret_cast<const floatx4 *>(red + 4);

        se_prev = se;
        se = se_next;
        // this pass's columns are stored at its own sample scale; the transition strip (the previous pass's last T-1
        // columns + copies of this pass's first T-1) at the smaller of the two passes' scales, where neither overflows
        cse = scaling != 0 ? 0 : se;
        const int csx = scaling != 0 ? 0 : ((p > 0 && se_prev < se) ? se_prev : se);
        cse_post = cse;
        csx_post = csx;
#include "fused_strip.inc"
        // ---- magnitude (zvabs/2 :329-333), scaling (SyllableDetector.swift:184-212),
        // statistic, f16 split, column -> LDS.  Result layout: column = frame f, register j of lane group g4 in
        // tile m = basis row 16m + 4*g4 + j; this lane holds bins 4*g4 + j (i = j) and 16 + 4*g4 + j (i = 4 + j).
        if (SPECT) {
            // spectrogram only: |X| (zvabs/2, :329-333) or |X|^2 (zvmags/4, :270-274) of this lane's 8 bins -> HBM
            const float inv = pow2f(-se - 13);
            const int64_t jf = e_b + (int64_t)kPass * p + fl;
            const bool live = jf < e_e;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float re = acc[i >> 2][i & 3] * inv, im = acc[2 + (i >> 2)][i & 3] * inv;
                const float pw = fmaf(re, re, im * im);
                const int bin = (i & 3) + 16 * (i >> 2) + 4 * g4;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d.spect_power ? pw : __builtin_amdgcn_sqrtf(pw)), spc_rs,
                                                      (live && bin < d.F) ? ((unsigned)jf * (unsigned)d.F + bin) * 4u : 0xFFFFFFFFu, 0, 0);
            }
        } else if (!(kom & 64)) {
#include "fused_mag.inc"
        } else if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.0f) colh[lane] = (_Float16)1.0f;
        SD_TICK(2)

        // ---- next pass: scale, split, -> LDS; then the loads of the pass after it start their way from HBM
        if (p + 1 < runs) {
            se_next = pass_scale(red0, red1);
            stage_pass(se_next);
            load_pass(p + 2);
        }
        SD_TICK(3)
        __syncthreads();          // columns of pass p and staged samples of pass p+1 are complete
        SD_TICK(4)
        if (STAMP) {                                          // one wait for all of this pass's ticks
            tsum[0] += tick[0] - tick[5];
#pragma unroll
            for (int i = 1; i < 5; i++) tsum[i] += tick[i] - tick[i - 1];
            tick[5] = tick[4];
        }
    }
    // ---- evaluation of the last pass
    if (!(kom & 128) && !SPECT) {
#pragma unroll
        for (int step = 0; step < 7; step++) post_step(step, runs - 1, cse_post, csx_post);
    }
    if (STAMP && tid == 0 && d.stamps)
        for (int i = 0; i < 8; i++) atomicAdd(&d.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + i], tsum[i]);
    if (STAMP && tid == 64 * (kWaves - 1) && d.stamps)       // the youngest wave's view of the same phases, slots 8..
        for (int i = 0; i < 8; i++) atomicAdd(&d.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + 8 + i], tsum[i]);
}

template <int KS, int TMAX, int NL, bool EXACT, bool SKEW, bool LEAN = false, bool STAMP = false, int KNOCK = 0, bool SPECT = false>
hipError_t launch_one(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t s_eff, int64_t E,
                      float *outputs, uint8_t *flags, hipStream_t stream)
{
    auto kern = fused_kernel<KS, TMAX, NL, EXACT, SKEW, LEAN, STAMP, KNOCK, SPECT>;
    hipError_t st = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, d.lds_total);
    if (st != hipSuccess) return st;
    const int64_t segs = (E + d.seg_evals - 1) / d.seg_evals;
    dim3 grid((unsigned)segs, (unsigned)C);
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), (size_t)d.lds_total, stream, d, samples, stride, s_eff, E, outputs, flags);
    return hipGetLastError();
}

}  // namespace

int fused_taps_max(int T) { return T <= 12 ? 12 : 0; }

hipError_t launch_fused(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t S, int64_t J,
                        int64_t E, float *outputs, uint8_t *flags, hipStream_t stream)
{
    if (E <= 0 || C <= 0) return hipSuccess;
    // the register-resident-basis kernel where it is instantiated (SYLDET_FUSED_CLASSIC=1 keeps this file's kernel: A/B
    // runs and the tests that hold the two against each other; read per call so that one process can do both)
    const bool classic = std::getenv("SYLDET_FUSED_CLASSIC") != nullptr;
    if ((!classic || !d.classic_ok) && !d.ko && fused_r_applicable(d) && (!d.stamps || fused_r_has_stamps() || !d.classic_ok)) return launch_fused_r(d, samples, stride, C, S, J, E, outputs, flags, stream);
    if (!d.classic_ok) return hipErrorInvalidValue;
    // one past the last sample an existing frame reads: frame J-1 covers [(J-1)*hop + gap, ... + W)
    const int64_t s_eff = (J - 1) * (int64_t)d.hop + d.gap + d.W;
    const bool skew = d.skew != 0;
    // the reference's example shape (W = 256, hop 132, timeRange 10) gets an instantiation with exact sizes
    if (d.KS == 8 && d.T == 10 && d.nload == 9 && skew) {      // the same shape at hop 128 (bank-spread staging)
        const bool lean = d.norm == 1 && d.scaling == 0 && d.n_layers == 2 && d.tf0 == 0 /* TanSig */ && d.tf1 == 2 /* PureLin */ &&
                          d.n_out == 1 && d.H <= 4 && d.n_out_fns <= 1;
        if (lean) return launch_one<8, 10, 9, true, true, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
        return launch_one<8, 10, 9, true, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
    }
    if (d.KS == 8 && d.T == 10 && d.nload == 9 && !skew) {
        const bool lean = d.norm == 1 && d.scaling == 0 && d.n_layers == 2 && d.tf0 == 0 /* TanSig */ && d.tf1 == 2 /* PureLin */ &&
                          d.n_out == 1 && d.H <= 4 && d.n_out_fns <= 1;
#ifdef SYLDET_KNOCKOUTS
#define SD_KO_CASE(m) case m: return launch_one<8, 10, 9, true, false, true, false, m>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
        if (lean) switch (d.ko) { SD_KO_CASE(1) SD_KO_CASE(2) SD_KO_CASE(4) SD_KO_CASE(16) SD_KO_CASE(64) SD_KO_CASE(128) SD_KO_CASE(144) SD_KO_CASE(256) SD_KO_CASE(254) SD_KO_CASE(507) SD_KO_CASE(511) default: break; }
#endif
        if (lean && d.stamps) return launch_one<8, 10, 9, true, false, true, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
        if (lean) return launch_one<8, 10, 9, true, false, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
        return launch_one<8, 10, 9, true, false>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
    }
    // everything else: runtime sizes, with the tap array (8 registers per tap) sized to the next of 4 / 8 / 12
#define SD_GENERIC(KS_, TM_)                                                                                               \
    return skew ? launch_one<KS_, TM_, kFusedMaxLoads, false, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream) \
                : launch_one<KS_, TM_, kFusedMaxLoads, false, false>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
    if (d.KS == 8) {
        if (d.T <= 4) { SD_GENERIC(8, 4) }
        if (d.T <= 8) { SD_GENERIC(8, 8) }
        SD_GENERIC(8, 12)
    }
    if (d.KS == 4) {
        if (d.T <= 4) { SD_GENERIC(4, 4) }
        if (d.T <= 8) { SD_GENERIC(4, 8) }
        SD_GENERIC(4, 12)
    }
#undef SD_GENERIC
    return hipErrorInvalidValue;
}

// The DFT front half alone: samples -> [C][J][F] columns (|X| or |X|^2) in HBM.  `d` is a plan built for timeRange 1.
hipError_t launch_fused_spectrogram(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t J, hipStream_t stream)
{
    if (J <= 0 || C <= 0) return hipSuccess;
    if ((uint64_t)J * (uint64_t)d.F * 4u >= 0xFFFFFFF0ull) return hipErrorInvalidValue;    // 32-bit byte offsets per channel
    const int64_t s_eff = (J - 1) * (int64_t)d.hop + d.gap + d.W;
    const bool skew = d.skew != 0;
#define SD_SPECT(KS_)                                                                                                        \
    return skew ? launch_one<KS_, 4, kFusedMaxLoads, false, true, false, false, 0, true>(d, samples, stride, C, s_eff, J, nullptr, nullptr, stream) \
                : launch_one<KS_, 4, kFusedMaxLoads, false, false, false, false, 0, true>(d, samples, stride, C, s_eff, J, nullptr, nullptr, stream);
    if (d.KS == 8) { SD_SPECT(8) }
    if (d.KS == 4) { SD_SPECT(4) }
#undef SD_SPECT
    return hipErrorInvalidValue;
}

}  // namespace sd
