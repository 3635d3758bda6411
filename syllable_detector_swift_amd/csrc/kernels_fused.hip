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
    // (In registers, 8 a tap -- except in the exact-size instantiation with bank-spread staging and the network as run-time
    // facts, which spills 91 registers that way and 40 when it fetches a tap's pair when it is due, three taps ahead, from the
    // table in memory (every workgroup reads the same few KB, which stay in the caches): 8 hidden units at hop 128 1.98 ms
    // against 3.30.  Measured the other way round elsewhere: without the padding 1.86 against 1.81, and the instantiations
    // with run-time sizes -- which spill 70 - 146 registers either way -- 7.4 ms against 4.1 at timeRange 12.)
    constexpr bool kAfrRegs = LEAN || SPECT || !(EXACT && SKEW);
    half8 afr[kAfrRegs ? TMAX : 1][2];
    const uint32x4 *afr_mem = reinterpret_cast<const uint32x4 *>(d.afrag) + lane;
#pragma unroll
    for (int t = 0; t < (kAfrRegs ? TMAX : 1); t++)
#pragma unroll
        for (int p = 0; p < 2; p++)
            afr[t][p] = (SPECT || !kAfrRegs) ? half8{} : as_half8(afr_mem[((t < T ? t : 0) * 2 + p) * 64]);
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
    // (status of the pass for the precision guard: 0 fine, 1 silent -- all samples zero, its columns are exact zeros --, 2 the
    // grid cannot hold it: an infinite sample, or a level above 2^113; a level below 2^-100 only loses headroom, which the
    // guard's own criterion sees)
    auto pass_scale = [&](floatx4 r0, floatx4 r1, int &status) {
        const float amax = fmaxf(fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[2], r0[3])), fmaxf(fmaxf(r1[0], r1[1]), fmaxf(r1[2], r1[3])));
        const int ex = (int)((__float_as_uint(amax) >> 23) & 0xffu);
        int e = 13 - ex + 127;
        status = __builtin_amdgcn_readfirstlane(amax > 0.0f ? ((ex == 255 || e < -100) ? 2 : 0) : 1);
        e = amax > 0.0f ? (e < -100 ? -100 : (e > 113 ? 113 : e)) : 0;       // (2^(-e - 13) must stay a normal number)
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
                // (quads past the pass are not staged: with exact sizes the nine quads of a thread are loaded whatever the hop,
                // and those past the pass came back as zeros from the descriptor's bounds check)
                if (EXACT ? i < d.nsmp : (k + 1 < nload || i < d.nsmp)) {
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
    int se_next, st_next;             // scale exponent and guard status of the pass staged most recently
    load_pass(0);
    max_partial();
    __syncthreads();
    se_next = pass_scale(*reinterpret_cast<const floatx4 *>(red), *reinterpret_cast<const floatx4 *>(red + 4), st_next);
    stage_pass(se_next);
    load_pass(1);                     // arrives during the first pass's matrix work
    int se = 0, cse = 0, se_prev = 0;   // sample / column scale exponents of this pass, sample scale of the one before
    int st = 1, st_prev = 1;            // their guard statuses
    // the window's sum of squares is kept for l2normalize and, for the precision guard, for linear columns without a normaliser
    const bool want_ss = norm == 1 || (norm == 0 && scaling == 0);
    bool badv = false;                  // this lane's evaluation of the pass failed the guard (kernels.hpp, FixItem)
    const bool guard_on = d.fix.counters != nullptr;
    unsigned long long tsum[16] = {0}, tick[8] = {0};
    if (STAMP) tick[5] = __builtin_amdgcn_s_memtime();
    __syncthreads();

    // ---- evaluation of one pass in eight steps, so that it can ride along with the NEXT pass's matrix work:
    // the first layer as a shifted GEMM over the column buffer (steps 0-2), the rest of the network in registers
    // (3-5), stores (6).  This wave finishes evaluation slots 16*wave .. +15 of the pass (slot q: e = e_b + 128 pp -
    // (T-1) + q, columns q .. q+T-1): result column = f, rows 4*g4 + j = hidden unit.
    floatx4 z = {0.0f, 0.0f, 0.0f, 0.0f}, z2 = {0.0f, 0.0f, 0.0f, 0.0f};
    float ssw = 1.0f;                                         // l2normalize: the window's sum of squares
    float alpha = 0.0f, beta = 0.0f, act[4] = {0.0f, 0.0f, 0.0f, 0.0f}, yv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    bool hit = false;
    constexpr int kAhead = 3;                                 // column fragments are fetched this many taps ahead
    uint32x4 bh_q[kAhead], bl_q[kAhead];
    uint32x4 ah_q[kAhead], al_q[kAhead];                     // (first-layer fragments of the taps ahead, where they are not resident)
    // Evaluation slot q < T-1 straddles two passes: its window is the transition strip [T-1 carried columns | copies of
    // this pass's first T-1 columns], kept at a scale both passes fit in; every other window reads the pass's own
    // columns at the pass's own scale.  Either way the window is T consecutive slots starting at `wslot`.
    const int wslot = fl < T - 1 ? fl : XS + fl - (T - 1);
    const _Float16 *bph = colh + wslot * kColStride + 8 * g4, *bpl = coll + wslot * kColStride + 8 * g4;
    auto gemm0_taps = [&](int t0, int t1) {
#pragma unroll
        for (int t = 0; t < TMAX; t++) {
            if (t >= t0 && t < t1 && t < T && !(kom & 16)) {
                const half8 h0 = as_half8(bh_q[t % kAhead]), l0 = as_half8(bl_q[t % kAhead]);
                const half8 a0 = kAfrRegs ? afr[kAfrRegs ? t : 0][0] : as_half8(ah_q[t % kAhead]);
                const half8 a1 = kAfrRegs ? afr[kAfrRegs ? t : 0][1] : as_half8(al_q[t % kAhead]);
                if (t + kAhead < T) {
                    bh_q[t % kAhead] = *reinterpret_cast<const uint32x4 *>(bph + (t + kAhead) * kColStride);
                    bl_q[t % kAhead] = *reinterpret_cast<const uint32x4 *>(bpl + (t + kAhead) * kColStride);
                    if (!kAfrRegs) {
                        // (the address goes through an opaque statement here: left alone the compiler hoists the fetches of
                        // all taps to the top of the evaluation and the fragments are resident again)
                        const uint32x4 *pm = afr_mem + ((t + kAhead) * 2) * 64;
                        asm volatile("" : "+v"(pm));
                        ah_q[t % kAhead] = pm[0];
                        al_q[t % kAhead] = pm[64];
                    }
                }
                z = mfma(a0, h0, z);                          // two accumulation chains: hi*hi on one,
                z2 = mfma(a0, l0, z2);                        // the cross terms on the other
                z2 = mfma(a1, h0, z2);
            }
        }
    };
    auto post_step = [&](int step, int pp, int cse_own, int cse_x, int st_own, int st_x) {
        const int cse_pp = fl < T - 1 ? cse_x : cse_own;      // column scale of this lane's window
        const int st_w = fl < T - 1 ? st_x : st_own;          // guard status of the passes its columns come from
        const int n0 = (T + 2) / 3, n1 = n0 + (T - n0 + 1) / 2;   // taps [0,n0), [n0,n1), [n1,T)
        if (step == 0) {
            z = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
            z2 = z;
#pragma unroll
            for (int t = 0; t < kAhead; t++)
                if (t < T && !(kom & 16)) {
                    bh_q[t] = *reinterpret_cast<const uint32x4 *>(bph + t * kColStride);
                    bl_q[t] = *reinterpret_cast<const uint32x4 *>(bpl + t * kColStride);
                    if (!kAfrRegs) {
                        ah_q[t] = afr_mem[(t * 2 + 0) * 64];
                        al_q[t] = afr_mem[(t * 2 + 1) * 64];
                    }
                }
            gemm0_taps(0, n0);
        } else if (step == 1) {
            gemm0_taps(n0, n1);
        } else if (step == 2) {
            gemm0_taps(n1, T);
            z += z2;
            if (want_ss) {                                    // sum of squares of the window = of its T frames (fp32, LDS)
                float acc_ss = 0.0f;
#pragma unroll
                for (int t = 0; t < TMAX; t++)
                    if (t < T) acc_ss += stat[wslot + t];
                ssw = acc_ss;
            }
        } else if (step == 3) {
            const float cs = scaling != 0 ? 1.0f : pow2f(cse_pp - d.col_shift);
            const float zs = d.w_unscale / cs;                // first-layer sums back to true units
            alpha = zs; beta = 0.0f;                          // layer-0 input = alpha * z + beta * rvec + bias0
            if (norm == 1) {                                  // L2Normalize, NeuralNet.swift:47-59
                // z and the per-frame sums of squares are both in column units: layer-0 input = W0 . v / |v|
                alpha = d.w_unscale * __builtin_amdgcn_rsqf(ssw);
            } else if (norm == 2) {                           // Normalize, :69-96
                float mn = INFINITY, mx = -INFINITY;
                for (int t = 0; t < T; t++) { mn = fminf(mn, stat[wslot + t]); mx = fmaxf(mx, stat[PS + wslot + t]); }
                const float range = mx - mn;
                if (range == 0.0f) { alpha = 0.0f; beta = -1.0f; }
                else { alpha = zs * 2.0f / range; beta = (0.0f - mn - mx) / range; }
            } else if (norm == 3) {                           // NormalizeStd, :105-108 (population sigma)
                float nn = 0.0f, mean = 0.0f, m2 = 0.0f;
                for (int t = 0; t < T; t++) {                 // pairwise-stable combination of per-frame (mean, M2)
                    const float nb = (float)d.F, tot = nn + nb, dlt = stat[wslot + t] - mean;
                    // (the two updates stay scalar: packed into v_pk_mul_f32 the compiler took the common factors from the high
                    // registers of src1 by operand selection -- the form tools/check_pk_opsel.py refuses, MEASUREMENTS R5.1)
                    float m2_step = dlt * dlt * nn * nb / tot;
                    asm volatile("" : "+v"(m2_step));
                    float mean_step = dlt * nb / tot;
                    asm volatile("" : "+v"(mean_step));
                    mean += mean_step;
                    m2 += stat[PS + wslot + t] + m2_step;
                    asm volatile("" : "+v"(mean), "+v"(m2));
                    nn = tot;
                }
                const float sd = sqrtf(m2 / (float)d.I);
                alpha = zs / sd;
                beta = -mean / sd;
                ssw = sd;                                     // (the guard's statistic)
            }
            // Precision guard: the window's statistic, on the grid its columns were stored on, against the grid's floor
            // times the network's sensitivity (fused_plan.cpp).  Exact zeros of silent passes are the reference's own values;
            // a pass the grid cannot hold condemns every window that touches it.  log / dB columns are not guarded beyond that.
            {
                const int64_t e = e_b + (int64_t)kPass * pp - (T - 1) + fl;
                const bool valid = e >= e_b && e < e_e && pp >= 0;
                float gthr = 0.0f, gstat = ssw;
                if (scaling == 0) {
                    if (norm == 1) gthr = d.guard_c;
                    else if (norm == 0) gthr = cse_pp < d.guard_se_abs_c ? d.guard_rel_c : 0.0f;
                    else gthr = d.guard_c_range;
                    if (norm == 2) {                          // (min / max and mean / M2 are kept in true units: onto the grid)
                        float mn = INFINITY, mx = -INFINITY;
                        for (int t = 0; t < T; t++) { mn = fminf(mn, stat[wslot + t]); mx = fmaxf(mx, stat[PS + wslot + t]); }
                        gstat = (mx - mn) * cs;
                    } else if (norm == 3) {
                        gstat = ssw * cs;
                    } else if (norm == 0) {                   // every column on its own: the quietest one of the window
                        float mn = INFINITY;
                        for (int t = 0; t < T; t++) mn = fminf(mn, stat[wslot + t]);
                        gstat = mn;
                    }
                } else gstat = 0.0f;
                badv = valid && (st_w == 2 || (st_w == 0 && !(gstat >= gthr)));
            }
        } else if (step == 4) {                               // rows past H (padding, statistic) contribute nothing
#pragma unroll
            for (int j = 0; j < 4; j++)
                act[j] = (LEAN || (4 * g4 + j) < H) ? transfer_fn(tf0, fmaf(alpha, z[j], fmaf(beta, c_rv[j], c_b0[j]))) : 0.0f;   // LEAN: padding rows meet zero weights
        } else if (step == 5) {
            const double *thr = reinterpret_cast<const double *>(cst + kCstThr);
            hit = false;
            if (n_layers == 2) {
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    if (o < n_out) {
                        float y = c_w1[o][0] * act[0];        // padding rows carry zero weights
                        y = fmaf(c_w1[o][1], act[1], y);
                        y = fmaf(c_w1[o][2], act[2], y);
                        y = fmaf(c_w1[o][3], act[3], y);
                        if (!LEAN && H > 4) {
                            y += __shfl_xor(y, 16, 64);
                            y += __shfl_xor(y, 32, 64);
                        }
                        y = transfer_fn(tf1, y + c_b1[o]);
                        if (LEAN) y = (y - lean_oa) / lean_og + lean_ob;
                        else
                            for (int kf = 0; kf < d.n_out_fns; kf++) {    // reverse maps, NeuralNet.swift:137-142 / :175-180
                                const float *op = cst + kCstOut + kf * (1 + 2 * n_out);
                                y = (y - op[0]) / op[1 + o] + op[1 + n_out + o];
                            }
                        yv[o] = y;
                        hit = hit | ((o == 0 || d.rule == 1) & ((double)y >= thr[o]));
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int h = 4 * g4 + j;
                    float y = act[j];
                    for (int kf = 0; kf < d.n_out_fns; kf++) {
                        const float *op = cst + kCstOut + kf * (1 + 2 * n_out);
                        y = (y - op[0]) / op[1 + (h < H ? h : 0)] + op[1 + n_out + (h < H ? h : 0)];
                    }
                    yv[j] = y;
                    if (h < H && (h == 0 || d.rule == 1)) hit = hit || ((double)y >= thr[h]);
                }
                int anyhit = hit ? 1 : 0;
                anyhit |= __shfl_xor(anyhit, 16, 64);
                anyhit |= __shfl_xor(anyhit, 32, 64);
                hit = anyhit != 0;
            }
        } else if (step == 6) {
            const int64_t e = e_b + (int64_t)kPass * pp - (T - 1) + fl;
            const bool valid = e >= e_b && e < e_e && pp >= 0 && !(kom & 8);
            const unsigned off = (unsigned)e;                 // E * n_out * 4 < 2^32 is checked by the launcher
            if (n_layers == 2) {
                const bool st = valid && g4 == 0;
#pragma unroll
                for (int o = 0; o < 4; o++)
                    if (o < n_out)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yv[o]), out_rs, st ? (off * n_out + o) * 4u : 0xFFFFFFFFu, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(hit ? 1 : 0), flg_rs, st ? off : 0xFFFFFFFFu, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yv[j]), out_rs,
                                                          (valid && (4 * g4 + j) < H) ? (off * n_out + 4 * g4 + j) * 4u : 0xFFFFFFFFu, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(hit ? 1 : 0), flg_rs, (valid && g4 == 0) ? off : 0xFFFFFFFFu, 0, 0);
            }
        }
    };
    int cse_post = 0, csx_post = 0;                           // column scales (own, transition strip) of the pass being evaluated
    int st_post = 1, stx_post = 1;                            // and the guard statuses of the passes behind them
    // evaluations of pass pp that failed the guard: this wave's 16 go to the work list as one item (rare; outside block M)
    auto push_bad = [&](int pp) {
        if (guard_on && __builtin_amdgcn_ballot_w64(badv) != 0ull) {
            int64_t lo = e_b + (int64_t)kPass * pp - (T - 1) + 16 * wave, hi = lo + 16;
            lo = lo < e_b ? e_b : lo;
            hi = hi > e_e ? e_e : hi;
            if (lane == 0 && hi > lo) {
                const unsigned slot = atomicAdd(d.fix.counters, 1u);
                if (slot < d.fix.capacity) d.fix.items[slot] = FixItem{c, (unsigned)lo, (int)(hi - lo), 0};
                else d.fix.counters[3] = 1u;
            }
        }
        badv = false;
    };
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
                    if (KS == 8) post_step(ks, p - 1, cse_post, csx_post, st_post, stx_post);
                    else { post_step(2 * ks, p - 1, cse_post, csx_post, st_post, stx_post); post_step(2 * ks + 1, p - 1, cse_post, csx_post, st_post, stx_post); }
                }
            }
        }
        // block-max partial of the next pass's samples (fetched a pass ago)
        if (p + 1 < runs) max_partial();
        SD_TICK(0)
        __syncthreads();          // all reads of the staged samples and of the columns are done; partial maxima are in
        SD_TICK(1)
        const floatx4 red0 = *reinterpret_cast<const floatx4 *>(red), red1 = *reinterpret_cast<const floatx4 *>(red + 4);
        if (!SPECT) push_bad(p - 1);

        se_prev = se;
        se = se_next;
        st_prev = st;
        st = st_next;
        // this pass's columns are stored at its own sample scale; the transition strip (the previous pass's last T-1
        // columns + copies of this pass's first T-1) at the smaller of the two passes' scales, where neither overflows
        cse = scaling != 0 ? 0 : se;
        const int csx = scaling != 0 ? 0 : ((p > 0 && se_prev < se) ? se_prev : se);
        cse_post = cse;
        csx_post = csx;
        st_post = st;
        stx_post = p > 0 ? ((st == 2 || st_prev == 2) ? 2 : ((st == 1 && st_prev == 1) ? 1 : 0)) : st;
        // ---- the previous pass's last T-1 columns -> the front of the transition strip (rescaled from their own scale);
        // done by the wave that overwrites their slots right after, so program order keeps the two apart
        if (!SPECT && p > 0 && wave == kWaves - 1 && !(kom & 32)) {
            const int dexp = scaling != 0 ? 0 : csx - se_prev;     // <= 0
            const int words = (T - 1) * (kColStride / 2);      // 32-bit words per array
            const int src = (XS + kPass - (T - 1)) * (kColStride / 2);
            // all reads first, then all writes: one LDS round trip instead of one per 64 words (this wave is the
            // last one through the phase, so its latency is the workgroup's)
            constexpr int kIt = TMAX > 1 ? (2 * (TMAX - 1) * (kColStride / 2) + 63) / 64 : 1;    // (timeRange 1: nothing is carried)
            unsigned u[kIt];
#pragma unroll
            for (int k = 0; k < kIt; k++) {
                const int i = lane + 64 * k;
                const bool hi_arr = i < words;
                const int w = hi_arr ? i : i - words;
                u[k] = i < 2 * words ? reinterpret_cast<const unsigned *>(hi_arr ? colh : coll)[src + w] : 0u;
            }
#pragma unroll
            for (int k = 0; k < kIt; k++) {
                const int i = lane + 64 * k;
                const bool hi_arr = i < words;
                const int w = hi_arr ? i : i - words;
                unsigned uu = u[k];
                if (dexp != 0) {
                    union { unsigned u; _Float16 h[2]; } x;
                    x.u = uu;
                    const float f0 = (float)x.h[0] * pow2f(dexp);
                    const float f1 = (float)x.h[1] * pow2f(dexp);
                    union { decltype(__builtin_amdgcn_cvt_pkrtz(0.f, 0.f)) h; unsigned u; } y;
                    y.h = __builtin_amdgcn_cvt_pkrtz(f0, f1);
                    uu = y.u;
                }
                if (i < 2 * words) reinterpret_cast<unsigned *>(hi_arr ? colh : coll)[w] = uu;
            }
            const int ssrc = XS + kPass - (T - 1);
            if (want_ss && lane < T - 1) stat[lane] = stat[ssrc + lane] * pow2f(2 * dexp);   // sums of squares of scaled columns
            if (norm >= 2 && lane < T - 1) {
                stat[lane] = stat[ssrc + lane];
                stat[PS + lane] = stat[PS + ssrc + lane];
            }
        }
        // ---- magnitude (zvabs/2 :329-333), scaling (SyllableDetector.swift:184-212),
        // statistic, f16 split, column -> LDS.  Result layout: column = frame f, register j of lane group g4 in
        // tile m = basis row 16m + 4*g4 + j; this lane holds bins 4*g4 + j (i = j) and 16 + 4*g4 + j (i = 4 + j).
        if (SPECT) {
            // spectrogram only: |X| (zvabs/2, :329-333) or |X|^2 (zvmags/4, :270-274) of this lane's 8 bins -> HBM
            const float inv = pow2f(-se - 13);
            const int64_t jf = e_b + (int64_t)kPass * p + fl;
            const bool live = jf < e_e;
            float gss = 0.0f;                                     // the frame's column sum of squares on the sample grid (guard)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                // (the square root is taken on the accumulator's scale, |acc| < 2^41: a recording at 1e30 must not overflow in
                // the squares of a magnitude that itself fits fp32)
                const float re = acc[i >> 2][i & 3], im = acc[2 + (i >> 2)][i & 3];
                const float pw = fmaf(re, re, im * im);
                gss += pw;
                const float mag = __builtin_amdgcn_sqrtf(pw) * inv;
                const int bin = (i & 3) + 16 * (i >> 2) + 4 * g4;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d.spect_power ? mag * mag : mag), spc_rs,
                                                      (live && bin < d.F) ? ((unsigned)jf * (unsigned)d.F + bin) * 4u : 0xFFFFFFFFu, 0, 0);
            }
            // Precision guard (kernels.hpp, FixItem): frames the pass's sample grid cannot hold -- an infinite sample in the
            // pass, or a column within reach of the grid's floor in a pass loud enough for that floor to matter -- are
            // recomputed from the samples.  16 frames of a wave are one work item.
            if (guard_on) {
                gss = xor32_sum(xor16_sum(gss)) * pow2f(2 * (-13 - d.col_shift));        // in column-grid units
                const bool bad = live && (st == 2 || (st == 0 && se < d.guard_se_abs_s && !(gss >= d.guard_spect)));
                if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
                    int64_t lo = e_b + (int64_t)kPass * p + 16 * wave, hi = lo + 16;
                    hi = hi > e_e ? e_e : hi;
                    if (lane == 0 && hi > lo) {
                        const unsigned slot = atomicAdd(d.fix.counters, 1u);
                        if (slot < d.fix.capacity) d.fix.items[slot] = FixItem{c, (unsigned)lo, (int)(hi - lo), 1};
                        else d.fix.counters[3] = 1u;
                    }
                }
            }
        } else if (!(kom & 64)) {
            // accumulators hold X * sx * 2^13.  Column scale (power of two; col_shift from the basis' largest row sum):
            // |X| * 2^(cse - shift) < 2^13; log/dB columns are stored unscaled.  For linear columns the two scales are applied together after the square root.
            const float inv = pow2f(-se - 13);
            const float cs = scaling != 0 ? 1.0f : pow2f(cse - d.col_shift);
            const int fh = d.F - 4 * g4;                          // cval[i] is a band bin iff (i&3) + 16(i>>2) < fh
            const bool plain = scaling == 0 && norm <= 1;         // linear |X| columns, no per-frame statistic on raw values
            float cval[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (plain) {
                    const float re = acc[i >> 2][i & 3], im = acc[2 + (i >> 2)][i & 3];
                    cval[i] = __builtin_amdgcn_sqrtf(fmaf(re, re, im * im)) * (inv * cs);   // |acc| < 2^40: no overflow
                } else {
                    const float re = acc[i >> 2][i & 3] * inv, im = acc[2 + (i >> 2)][i & 3] * inv;
                    const float pw = fmaf(re, re, im * im);
                    cval[i] = __builtin_amdgcn_sqrtf(pw);
                }
            }
            if (scaling != 0) {
                const float kk = scaling == 1 ? 0.6931471805599453f : 6.020599913279624f;   // ln 2, 20 log10 2
#pragma unroll
                for (int i = 0; i < 8; i++) cval[i] = kk * __builtin_amdgcn_logf(cval[i]);  // v_log_f32 = log2
#pragma unroll
                for (int i = 0; i < 8; i++) cval[i] = ((i & 3) + 16 * (i >> 2)) < fh ? cval[i] : 0.0f;   // log(0) rows
            }                                                     // (linear: basis rows past F are zero, so are their |X|)
            const int slot = XS + fl;                             // own column; frames fl < T-1 also feed the transition strip
            const int xslot = (T - 1) + fl;
            if (norm == 2) {
                float st0 = INFINITY, st1 = -INFINITY;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const bool valid = ((i & 3) + 16 * (i >> 2)) < fh;
                    st0 = valid ? fminf(st0, cval[i]) : st0;
                    st1 = valid ? fmaxf(st1, cval[i]) : st1;
                }
                st0 = fminf(st0, __shfl_xor(st0, 16, 64)); st0 = fminf(st0, __shfl_xor(st0, 32, 64));
                st1 = fmaxf(st1, __shfl_xor(st1, 16, 64)); st1 = fmaxf(st1, __shfl_xor(st1, 32, 64));
                if (g4 == 0) {
                    stat[slot] = st0; stat[PS + slot] = st1;
                    if (fl < T - 1) { stat[xslot] = st0; stat[PS + xslot] = st1; }
                }
            } else if (norm == 3) {
                float st0 = 0.0f, st1 = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; i++) st0 += cval[i];
                st0 += __shfl_xor(st0, 16, 64); st0 += __shfl_xor(st0, 32, 64);
                st0 = st0 / (float)d.F;                           // mean of this frame's column
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float dlt = cval[i] - st0;
                    st1 = ((i & 3) + 16 * (i >> 2)) < fh ? fmaf(dlt, dlt, st1) : st1;   // M2 of this frame's column
                }
                st1 += __shfl_xor(st1, 16, 64); st1 += __shfl_xor(st1, 32, 64);
                if (g4 == 0) {
                    stat[slot] = st0; stat[PS + slot] = st1;
                    if (fl < T - 1) { stat[xslot] = st0; stat[PS + xslot] = st1; }
                }
            }
            if (!plain) {
#pragma unroll
                for (int i = 0; i < 8; i++) cval[i] *= cs;
            }
            if (want_ss) {
                // sum of squares of the (scaled) column, fp32, one value per frame next to the columns
                float ss = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; i++) ss = fmaf(cval[i], cval[i], ss);
                ss = xor32_sum(xor16_sum(ss));
                if (g4 == 0) {
                    stat[slot] = ss;
                    if (fl < T - 1) stat[xslot] = ss * pow2f(2 * (csx - cse));
                }
            }
            {
                _Float16 *ph = colh + slot * kColStride + 4 * g4, *pl = coll + slot * kColStride + 4 * g4;
#pragma unroll
                for (int m = 0; m < 2; m++) {                     // bins 16m + 4*g4 .. +3: four consecutive halves
                    unsigned h0, l0, h1, l1;
                    split_pair_scaled(cval[4 * m], cval[4 * m + 1], 1.0f, h0, l0);
                    split_pair_scaled(cval[4 * m + 2], cval[4 * m + 3], 1.0f, h1, l1);
                    uint32x2 uh = {h0, h1}, ul = {l0, l1};
                    *reinterpret_cast<uint32x2 *>(ph + 16 * m) = uh;
                    *reinterpret_cast<uint32x2 *>(pl + 16 * m) = ul;
                }
            }
            if (wave == 0 && fl < T - 1) {                        // copies for the transition strip, at its scale
                const float xs = pow2f(csx - cse);                // <= 1: a much quieter pass may underflow here, next to
                _Float16 *ph = colh + xslot * kColStride + 4 * g4, *pl = coll + xslot * kColStride + 4 * g4;   // columns 2^|.| louder
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    unsigned h0, l0, h1, l1;
                    split_pair_scaled(cval[4 * m], cval[4 * m + 1], xs, h0, l0);
                    split_pair_scaled(cval[4 * m + 2], cval[4 * m + 3], xs, h1, l1);
                    uint32x2 uh = {h0, h1}, ul = {l0, l1};
                    *reinterpret_cast<uint32x2 *>(ph + 16 * m) = uh;
                    *reinterpret_cast<uint32x2 *>(pl + 16 * m) = ul;
                }
            }
        } else if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.0f) colh[lane] = (_Float16)1.0f;
        SD_TICK(2)

        // ---- next pass: scale, split, -> LDS; then the loads of the pass after it start their way from HBM
        if (p + 1 < runs) {
            se_next = pass_scale(red0, red1, st_next);
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
        for (int step = 0; step < 7; step++) post_step(step, runs - 1, cse_post, csx_post, st_post, stx_post);
        push_bad(runs - 1);
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

// Which of the three fused kernels runs a plan: 2 the symmetric-fold kernel (kernels_fused_s.hip) where the shape is of its
// class, 1 the register-resident-basis kernel (kernels_fused_r.hip), 0 this file's 8-wave kernel.  A handle created under
// SYLDET_FUSED_NOFOLD=1 / SYLDET_FUSED_CLASSIC=1 keeps the older kernels (A/B runs, and the tests that hold them against
// each other); the diagnostic instantiations (stamps, knock-outs) exist for the older two only.
int fused_choice(const FusedDesc &d, int64_t J)
{
    const bool classic = d.force_classic != 0;
    const int64_t s_eff = (J - 1) * (int64_t)d.hop + d.gap + d.W;
    // (hops that are multiples of 64 put every frame of a tile on the same LDS banks; the register-resident-basis kernel stages
    // them with padding and is the faster one there for 192- and 256-sample windows: 2.13 against 3.08 ms at hop 64, 1.23 against
    // 1.63 at hop 128 -- MEASUREMENTS R3.4)
    const bool banked = d.hop % 64 == 0 && d.W > 128 && !d.s_padp && fused_r_applicable(d) && (!classic || !d.classic_ok);
    if (!classic && !d.no_fold && !d.ko && (!d.stamps || fused_s_has_stamps()) && !banked && fused_s_applicable(d) && s_eff * 4 < 0x7fffffffll) return 2;
    if ((!classic || !d.classic_ok) && !d.ko && fused_r_applicable(d) && (!d.stamps || fused_r_has_stamps() || !d.classic_ok)) return 1;
    return 0;
}

hipError_t launch_fused(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t S, int64_t J,
                        int64_t E, float *outputs, uint8_t *flags, hipStream_t stream)
{
    if (E <= 0 || C <= 0) return hipSuccess;
    const int choice = fused_choice(d, J);
    if (choice == 2) return launch_fused_s(d, samples, stride, C, S, J, E, outputs, flags, stream);
    if (choice == 1) return launch_fused_r(d, samples, stride, C, S, J, E, outputs, flags, stream);
    if (!d.classic_ok) return hipErrorInvalidValue;
    // one past the last sample an existing frame reads: frame J-1 covers [(J-1)*hop + gap, ... + W)
    const int64_t s_eff = (J - 1) * (int64_t)d.hop + d.gap + d.W;
    const bool skew = d.skew != 0;
    // the reference's example shape (W = 256, hop 132, timeRange 10) gets an instantiation with exact sizes
    // (every other timeRange at any hop up to 140, with the network as run-time facts: exact sizes too -- the instantiations
    // with run-time sizes spill 70 - 146 registers and run at half the speed)
#define SD_EXACT(KS_, T_)                                                                                                    \
    if (d.KS == KS_ && d.T == T_ && d.nload <= 9 && !(KS_ == 8 && T_ == 10)) {                                               \
        if (skew) return launch_one<KS_, T_, 9, true, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);        \
        return launch_one<KS_, T_, 9, true, false>(d, samples, stride, C, s_eff, E, outputs, flags, stream);                 \
    }
#define SD_EXACT_ALL(KS_) SD_EXACT(KS_, 1) SD_EXACT(KS_, 2) SD_EXACT(KS_, 3) SD_EXACT(KS_, 4) SD_EXACT(KS_, 5) SD_EXACT(KS_, 6) \
    SD_EXACT(KS_, 7) SD_EXACT(KS_, 8) SD_EXACT(KS_, 9) SD_EXACT(KS_, 10) SD_EXACT(KS_, 11) SD_EXACT(KS_, 12)
    SD_EXACT_ALL(8)
    SD_EXACT_ALL(4)
#undef SD_EXACT_ALL
#undef SD_EXACT
    if (d.KS == 8 && d.T == 10 && d.nload <= 9 && skew) {      // the same shape at hop 128 (bank-spread staging)
        const bool lean = d.norm == 1 && d.scaling == 0 && d.n_layers == 2 && d.tf0 == 0 /* TanSig */ && d.tf1 == 2 /* PureLin */ &&
                          d.n_out == 1 && d.H <= 4 && d.n_out_fns <= 1;
        if (lean) return launch_one<8, 10, 9, true, true, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
        return launch_one<8, 10, 9, true, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
    }
    if (d.KS == 8 && d.T == 10 && d.nload <= 9 && !skew) {
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
