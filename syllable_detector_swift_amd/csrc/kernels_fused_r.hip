// kernels_fused_r.hip -- the fused engine with the DFT basis resident in registers.
//
// Same path, same arithmetic and same tables as kernels_fused.hip (reference, root relative:
//   extractPower          Common/CircularShortTimeFourierTransform.swift:280-337
//   processFourierData    Common/SyllableDetector.swift:134-151
//   processNewValue       Common/SyllableDetector.swift:153-217
//   NeuralNet.apply       Common/NeuralNet.swift:294-326, :366-377
//   lastDetected          Common/SyllableDetector.swift:27-31),
// a different use of the CU.  A gfx950 wave that is alone on its SIMD owns 512 registers: the 64 KB of f16 hi/lo
// basis fragments (A operands of the windowed band-limited DFT) are exactly 256 of them, so
//   * one workgroup = 4 waves, one per SIMD, 16 frames each: 64-frame passes;
//   * the basis never travels through LDS again -- a k-step fetches two B fragments (staged samples, hi + lo) for
//     its 12 MFMAs instead of ten fragments, which a lone wave can do under the MFMAs;
//   * the 64 KB of LDS the basis used to take hold a second staged-sample buffer: pass p+1 is scaled, split and
//     staged, and pass p+2's loads are issued, INSIDE the matrix block of pass p, together with the evaluation of
//     pass p-1 -- one branch-free scheduling region in which vector, LDS and memory instructions ride between
//     the MFMAs of the same wave.  Outside it only the magnitudes, the transition strip and the block maximum
//     are left.
//
//   block M(p):  DFT(p) from staged buffer p&1  ||  evaluation of pass p-1 (columns)  ||  stage pass p+1 into the
//                other buffer  ||  reload the staging registers with pass p+2
//   barrier      (columns no longer read)
//   transition strip, |X|(p) -> columns, block-max partial of pass p+2
//   barrier      (columns, staged pass p+1 and the partial maxima are complete)
//
// gfx950 only.  wave = 64.

#include "fused_common.hpp"

namespace sd {

namespace {

using namespace fused_dev;

constexpr int kBlock = kFusedRBlock;           // 256 threads = 4 waves, one per SIMD
constexpr int kWaves = kBlock / 64;
constexpr int kPass = kFusedRTileFrames;       // 64 frames per pass = 16 per wave
constexpr int kColStride = kFusedColStride;

// KS: k-steps of 32 samples (the basis takes 32 KS registers); TMAX: taps the first-layer fragment array is sized
// for; NL: staging quads per thread (all NL are always loaded and staged: quads past the pass come back as zeros
// from the descriptor's bounds check and land in LDS words no frame reads); EXACT: timeRange == TMAX; SKEW: staged
// samples carry bank-spreading padding; LEAN: the reference's example configuration class as a compile-time fact.
template <int KS, int TMAX, int NL, bool EXACT, bool SKEW, bool LEAN, bool STAMP>
__global__ void __launch_bounds__(kBlock, 1)
fused_r_kernel(const FusedDesc d, const float *__restrict__ samples, int64_t stride, int64_t s_eff, int64_t E,
               float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *red = reinterpret_cast<float *>(smem + d.r_lds_red);      // [4 waves] block-max partials
    float *cst = reinterpret_cast<float *>(smem + d.r_lds_cst);
    constexpr int kom = 0;                // the shared blocks' diagnostic switches: none here
    constexpr bool SPECT = false;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = lane & 15;          // frame (DFT) / evaluation (first layer) column inside the wave's tile
    const int g4 = lane >> 4;         // k block 8*g4..8*g4+7 of an operand; rows 4*g4..4*g4+3 of a result
    const int c = blockIdx.y;
    const int64_t e_b = (int64_t)blockIdx.x * d.r_seg_evals;
    if (e_b >= E) return;
    const int64_t e_e = (e_b + d.r_seg_evals < E) ? e_b + d.r_seg_evals : E;
    const float *row = samples + (int64_t)c * stride;
    const int PS = d.r_ps, T = EXACT ? TMAX : d.T, H = d.H;    // PS: column slots = 2 (T - 1) transition slots + 64
    const int XS = 2 * (T - 1);                                 // the pass's own columns start at slot XS
    const int norm = LEAN ? 1 : d.norm, scaling = LEAN ? 0 : d.scaling;
    const int n_layers = LEAN ? 2 : d.n_layers, n_out = LEAN ? 1 : d.n_out, tf0 = LEAN ? 0 : d.tf0, tf1 = LEAN ? 2 : d.tf1;
    const int fl = 16 * wave + f;     // this lane's frame / evaluation slot inside the pass
    const int runs = d.r_runs;

    // staged samples: [buffer 0 hi | buffer 0 lo | buffer 1 hi | buffer 1 lo], r_smp_stride halves each
    _Float16 *smp0 = reinterpret_cast<_Float16 *>(smem + d.r_lds_smp);
    const int buf_halves = 2 * d.r_smp_stride;
    _Float16 *colh = reinterpret_cast<_Float16 *>(smem + d.r_lds_colh);   // [PS][kColStride] |X| columns, hi parts
    _Float16 *coll = reinterpret_cast<_Float16 *>(smem + d.r_lds_coll);   //                               lo parts
    float *stat = reinterpret_cast<float *>(smem + d.r_lds_stat);          // [2][PS] per-frame statistics

    // ---- once per workgroup: constants
    if (tid < 16) reinterpret_cast<double *>(cst + kCstThr)[tid] = tid < n_out ? d.thresholds[tid] : 0.0;
    for (int i = tid; i < d.n_out_fns * (1 + 2 * n_out); i += kBlock) cst[kCstOut + i] = d.out_params[i];
    // the DFT basis: A-operand fragments [k-step][re 0-15, re 16-31, im 0-15, im 16-31][hi, lo], one quad per lane each
    uint32x4 a[KS * 8];
#pragma unroll
    for (int i = 0; i < KS * 8; i++) a[i] = reinterpret_cast<const uint32x4 *>(d.dfrag)[i * 64 + lane];
    // first-layer fragments, one (hi, lo) pair per tap
    half8 afr[TMAX][2];
#pragma unroll
    for (int t = 0; t < TMAX; t++)
#pragma unroll
        for (int p = 0; p < 2; p++)
            afr[t][p] = as_half8(reinterpret_cast<const uint32x4 *>(d.afrag)[((t < T ? t : 0) * 2 + p) * 64 + lane]);
    float c_b0[4], c_rv[4], c_w1[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int h = 4 * g4 + j;
        c_b0[j] = h < H ? d.bias0[h] : 0.0f;
        c_rv[j] = h < H ? d.rvec[h] : 0.0f;
#pragma unroll
        for (int o = 0; o < 4; o++) c_w1[o][j] = (n_layers == 2 && h < H && o < n_out) ? d.w1[o * H + h] : 0.0f;
    }
    float c_b1[4];
#pragma unroll
    for (int o = 0; o < 4; o++) c_b1[o] = (n_layers == 2 && o < n_out) ? d.b1[o] : 0.0f;

    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(
        outputs ? outputs + (int64_t)c * E * n_out : nullptr, 0, outputs ? (int)(E * n_out * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t flg_rs = __builtin_amdgcn_make_buffer_rsrc(flags ? flags + (int64_t)c * E : nullptr, 0, flags ? (int)E : 0, 0x00020000);
    float lean_oa = 0.0f, lean_og = 1.0f, lean_ob = 0.0f;
    if (LEAN && d.n_out_fns == 1) { lean_oa = d.out_params[0]; lean_og = d.out_params[1]; lean_ob = d.out_params[2]; }

    // this lane's frame in a staged buffer, and where k-step ks of lane group g4 starts inside it (see kernels_fused.hip)
    const int foff = fl * (d.hop + (SKEW ? d.skew : 0)) + (SKEW ? 0 : 8 * KS * g4);
    int ko[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) ko[ks] = SKEW ? d.koff[ks * 4 + g4] : 8 * ks;

    // raw samples of one pass: quads 4*(tid + 256 k), k < NL, through a bounds-checked descriptor
    uint32x4 v[NL];
    auto pass_rsrc = [&](int p) {
        return tile_rsrc(row, (e_b + (int64_t)kPass * p) * d.hop + d.gap, p < runs ? s_eff : 0, d.r_nsmp);
    };
    auto max_partial = [&]() {
        float amax = 0.0f;
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const floatx4 q = as_floatx4(v[k]);
            amax = absmax3(absmax3(amax, q[0], q[1]), q[2], q[3]);
        }
        amax = wave_max_nonneg(amax);
        if (lane == 0) red[wave] = amax;
    };
    auto pass_scale = [&]() {
        const floatx4 r0 = *reinterpret_cast<const floatx4 *>(red);
        const float amax = fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[2], r0[3]));
        int e = 13 - (int)((__float_as_uint(amax) >> 23) & 0xffu) + 127;
        e = amax > 0.0f ? (e < -100 ? -100 : (e > 100 ? 100 : e)) : 0;
        return __builtin_amdgcn_readfirstlane(e);
    };
    // where this thread's quad k lands in a staged buffer (halves)
    int spos[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int i = 4 * (tid + kBlock * k);
        spos[k] = SKEW ? i + d.skew * (int)__umulhi((unsigned)i, d.hop_magic) : i;
    }
    // quad k: scale, split into f16 hi + lo, -> staged buffer `wh` (lo array r_smp_stride halves further)
    auto stage_quad = [&](int k, float sx, _Float16 *wh) {
        const floatx4 q = as_floatx4(v[k]);
        unsigned h0, l0, h1, l1;
        split_pair_scaled(q[0], q[1], sx, h0, l0);
        split_pair_scaled(q[2], q[3], sx, h1, l1);
        uint32x2 uh = {h0, h1}, ul = {l0, l1};
        _Float16 *ph = wh + (SKEW ? spos[k] : 4 * tid + 4 * kBlock * k);
        *reinterpret_cast<uint32x2 *>(ph) = uh;
        *reinterpret_cast<uint32x2 *>(ph + d.r_smp_stride) = ul;
    };

    // ---- prologue: pass 0 staged, pass 1 in the staging registers with its block maximum published
    int se, se_prev = 0;                  // sample scale exponents of the pass in the matrix block, and of the one before
    {
        const __amdgpu_buffer_rsrc_t rs = pass_rsrc(0);
#pragma unroll
        for (int k = 0; k < NL; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * tid + 16 * kBlock * k, 0, 0);
        max_partial();
        __syncthreads();
        se = pass_scale();
#pragma unroll
        for (int k = 0; k < NL; k++) stage_quad(k, pow2f(se), smp0);
        const __amdgpu_buffer_rsrc_t rs1 = pass_rsrc(1);
#pragma unroll
        for (int k = 0; k < NL; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs1, 16 * tid + 16 * kBlock * k, 0, 0);
        __syncthreads();                  // every wave has read the partial maxima of pass 0
        max_partial();
        __syncthreads();
    }
    int cse = 0;
    // diagnostic instantiation only (SYLDET_FUSED_STAMPS=1): s_memtime at the phase boundaries of every pass
    unsigned long long tsum[8] = {0}, tick[8] = {0};
#define SD_RTICK(slot)                                                                     \
    if (STAMP) {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        tick[slot] = __builtin_amdgcn_s_memtime();                                         \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    }
    SD_RTICK(5)

#include "fused_eval.inc"

    // The evaluation, re-cut for a lone wave: one slot between every two groups of four DFT MFMAs (3 per k-step).  A
    // column fragment is fetched two slots before the MFMAs that use it, the window's statistics two slots before
    // they are summed -- no LDS round trip is waited for in the slot that issues it.
    // Its MFMAs are ordered assembly statements like the DFT's (three accumulation chains, so that consecutive uses of
    // one accumulator are a slot apart), and the statistics pass through an empty ordered statement where they are
    // consumed: otherwise the compiler moves the consumers up to the fetches and waits there.
    uint32x4 cq_h[3], cq_l[3];
    float wst[TMAX];
    floatx4 z3 = {0.0f, 0.0f, 0.0f, 0.0f};
    auto eval_slot = [&](int s, int pp, int cse_own, int cse_x) {
        if (s < T) {                                              // fetch tap s
            cq_h[s % 3] = *reinterpret_cast<const uint32x4 *>(bph + s * kColStride);
            cq_l[s % 3] = *reinterpret_cast<const uint32x4 *>(bpl + s * kColStride);
        }
        if (s >= 2 && s - 2 < T) {                                // multiply tap s - 2: hi*hi, hi*lo, lo*hi
            const int t = s - 2;
            if (t == 0) {
                asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(z) : "v"(afr[t][0]), "v"(cq_h[t % 3]));
                asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(z2) : "v"(afr[t][0]), "v"(cq_l[t % 3]));
                asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(z3) : "v"(afr[t][1]), "v"(cq_h[t % 3]));
            } else {
                asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(z) : "v"(afr[t][0]), "v"(cq_h[t % 3]));
                asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(z2) : "v"(afr[t][0]), "v"(cq_l[t % 3]));
                asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(z3) : "v"(afr[t][1]), "v"(cq_h[t % 3]));
            }
        }
        if (s == T && norm == 1) {
#pragma unroll
            for (int t = 0; t < TMAX; t++)
                if (t < T) wst[t] = stat[wslot + t];
        }
        if (s == T + 2) {
            asm volatile("s_nop 7" : "+v"(z), "+v"(z2), "+v"(z3));   // (the last tap's MFMAs are a slot back: long done)
            z += z2 + z3;
            if (norm == 1) {
                float acc_ss = 0.0f;
#pragma unroll
                for (int t = 0; t < TMAX; t++)
                    if (t < T) {
                        asm volatile("" : "+v"(wst[t]));
                        acc_ss += wst[t];
                    }
                ssw = acc_ss;
            }
        }
        if (s >= T + 3 && s <= T + 6) post_step(s - T, pp, cse_own, cse_x);
    };

    int cse_post = 0, csx_post = 0;       // column scales (own, transition strip) of the pass being evaluated
    for (int p = 0; p < runs; p++) {
        // ================= block M: DFT(p)  ||  evaluation of pass p-1  ||  stage pass p+1  ||  reload for pass p+2
        const int se_next = pass_scale();                             // pass p+1 (its partial maxima are in)
        const float sx_next = pow2f(se_next);
        const _Float16 *fph = smp0 + (p & 1) * buf_halves + foff, *fpl = fph + d.r_smp_stride;
        _Float16 *wh = smp0 + ((p + 1) & 1) * buf_halves;
        const __amdgpu_buffer_rsrc_t rs2 = pass_rsrc(p + 2);
        // The DFT's matrix instructions are written as ordered assembly statements for two reasons.  Register files: the
        // basis is only ever an A operand, so it lives in the 256 accumulation registers (the matrix pipe reads A operands
        // from there directly) and leaves the 256 architectural registers to everything else -- left to itself the
        // allocator does the opposite and then serialises the block to relieve the pressure.  Order: a lone wave has
        // nobody to hide an LDS round trip behind, so every LDS / memory instruction of the block has a fixed place
        // between two groups of four MFMAs (the compiler keeps memory instructions on their side of such a statement);
        // vector instructions and the evaluation's own MFMAs move freely.  Hazards the compiler cannot see: the
        // accumulators are written by nothing else inside the block, consecutive uses of one accumulator are three
        // MFMAs apart, s_nops follow the last group before vector code reads the results, and two wait states in
        // front of every MFMA (an s_nop and the ordered vector instruction behind the previous MFMA; two s_nops in
        // the evaluation) cover a v_accvgpr_write / v_accvgpr_read of an operand placed right in front of it (the
        // allocator parks a few quads in the other register file when it runs short).
        floatx4 acc[4];
        {
            half8 bh = lds_half8(fph + ko[0]), bl = lds_half8(fpl + ko[0]);
            // A lone wave issues in order: while it waits at an MFMA for the matrix pipe it issues nothing else, so vector
            // work hides under the matrix work only if it sits BETWEEN the MFMAs in program order -- about three vector
            // instructions fit under each.  The next pass's staging (scale, f16 hi/lo split, two LDS writes, reload of the
            // quad with pass p+2) is therefore cut into single instructions, ordered statements like the MFMAs, one after
            // each DFT MFMA.
            unsigned mh0 = 0, ml0 = 0, mh1 = 0, ml1 = 0;
            auto micro = [&](int i) {
                const int k = i / 11, j = i % 11;
                if (k >= NL) return;
#ifndef SYLDET_R_NOSTAGE
                const floatx4 q = as_floatx4(v[k]);
                if (j == 0) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(mh0) : "v"(q[0]), "v"(sx_next));
                if (j == 1) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(mh0) : "v"(q[1]), "v"(sx_next));
                if (j == 2) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ml0) : "v"(q[0]), "v"(sx_next), "v"(mh0));
                if (j == 3) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(ml0) : "v"(q[1]), "v"(sx_next), "v"(mh0));
                if (j == 4) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(mh1) : "v"(q[2]), "v"(sx_next));
                if (j == 5) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(mh1) : "v"(q[3]), "v"(sx_next));
                if (j == 6) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ml1) : "v"(q[2]), "v"(sx_next), "v"(mh1));
                if (j == 7) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(ml1) : "v"(q[3]), "v"(sx_next), "v"(mh1));
                _Float16 *ph = wh + (SKEW ? spos[k] : 4 * tid + 4 * kBlock * k);
                if (j == 8) { uint32x2 uh = {mh0, mh1}; *reinterpret_cast<uint32x2 *>(ph) = uh; }
                if (j == 9) { uint32x2 ul = {ml0, ml1}; *reinterpret_cast<uint32x2 *>(ph + d.r_smp_stride) = ul; }
#endif
#ifndef SYLDET_R_NOLOAD
                if (j == 10) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs2, 16 * tid + 16 * kBlock * k, 0, 0);
#endif
            };
            auto slot = [&](int sl) {
#ifndef SYLDET_R_NOEVAL
                eval_slot(sl, p - 1, cse_post, csx_post);
#endif
            };
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const half8 cbh = bh, cbl = bl;
                if (ks + 1 < KS) {                                    // B fragments of the next k-step: a whole k-step ahead
                    bh = lds_half8(fph + ko[ks + 1]);
                    bl = lds_half8(fpl + ko[ks + 1]);
                }
                slot(3 * ks);
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    if (ks == 0) asm volatile("s_nop 0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc[m]) : "a"(a[ks * 8 + 2 * m]), "v"(cbh));
                    else asm volatile("s_nop 0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m]) : "a"(a[ks * 8 + 2 * m]), "v"(cbh));
                    micro(12 * ks + m);
                }
                slot(3 * ks + 1);
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    asm volatile("s_nop 0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m]) : "a"(a[ks * 8 + 2 * m]), "v"(cbl));
                    micro(12 * ks + 4 + m);
                }
                slot(3 * ks + 2);
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    asm volatile("s_nop 0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m]) : "a"(a[ks * 8 + 2 * m + 1]), "v"(cbh));
                    micro(12 * ks + 8 + m);
                }
            }
#pragma unroll
            for (int i = 12 * KS; i < 11 * NL; i++) micro(i);         // what is left of the last quad
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        }
        SD_RTICK(0)
        __syncthreads();          // all reads of the columns are done
        SD_RTICK(1)
        // this pass's columns are stored at its own sample scale; the transition strip (the previous pass's last T-1
        // columns + copies of this pass's first T-1) at the smaller of the two passes' scales, where neither overflows
        cse = scaling != 0 ? 0 : se;
        const int csx = scaling != 0 ? 0 : ((p > 0 && se_prev < se) ? se_prev : se);
        cse_post = cse;
        csx_post = csx;
#ifndef SYLDET_R_NOMAG
#include "fused_strip.inc"
        {
#include "fused_mag.inc"
        }
#endif
        SD_RTICK(2)
#ifndef SYLDET_R_NOMAX
        max_partial();
#endif
        SD_RTICK(3)            // pass p+2 (zeros past the segment)
        se_prev = se;
        se = se_next;
        __syncthreads();          // columns of pass p, staged samples of pass p+1 and the partial maxima are complete
        SD_RTICK(4)
        if (STAMP) {
            tsum[0] += tick[0] - tick[5];
#pragma unroll
            for (int i = 1; i < 5; i++) tsum[i] += tick[i] - tick[i - 1];
            tick[5] = tick[4];
        }
    }
    // ---- evaluation of the last pass
#pragma unroll
    for (int sl = 0; sl < 24; sl++) eval_slot(sl, runs - 1, cse_post, csx_post);
    if (STAMP && (tid == 0 || tid == 64 * (kWaves - 1)) && d.stamps)       // wave 0's view in slots 0-7, the last wave's in 8-15
        for (int i = 0; i < 8; i++) atomicAdd(&d.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (tid ? 8 : 0) + i], tsum[i]);
#undef SD_RTICK
}

template <int KS, int TMAX, int NL, bool EXACT, bool SKEW, bool LEAN = false, bool STAMP = false>
hipError_t launch_one(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t s_eff, int64_t E,
                      float *outputs, uint8_t *flags, hipStream_t stream)
{
    auto kern = fused_r_kernel<KS, TMAX, NL, EXACT, SKEW, LEAN, STAMP>;
    hipError_t st = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, d.r_lds_total);
    if (st != hipSuccess) return st;
    const int64_t segs = (E + d.r_seg_evals - 1) / d.r_seg_evals;
    dim3 grid((unsigned)segs, (unsigned)C);
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), (size_t)d.r_lds_total, stream, d, samples, stride, s_eff, E, outputs, flags);
    return hipGetLastError();
}

}  // namespace

// Shapes this kernel is instantiated for: 256-sample windows (8 k-steps), 9 staging quads per thread (hops 121..140:
// the reference's 132 and the 128 variant), timeRange 10.  Everything else stays on kernels_fused.hip's kernel.
bool fused_r_applicable(const FusedDesc &d)
{
    const bool lean = d.norm == 1 && d.scaling == 0 && d.n_layers == 2 && d.tf0 == 0 /* TanSig */ && d.tf1 == 2 /* PureLin */ &&
                      d.n_out == 1 && d.H <= 4 && d.n_out_fns <= 1;
    return d.r_ok && d.KS == 8 && d.T == 10 && d.r_nload == 9 && lean;
}

hipError_t launch_fused_r(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t S, int64_t J,
                          int64_t E, float *outputs, uint8_t *flags, hipStream_t stream)
{
    (void)S;
    if (E <= 0 || C <= 0) return hipSuccess;
    const int64_t s_eff = (J - 1) * (int64_t)d.hop + d.gap + d.W;
    if (d.stamps && d.skew == 0) return launch_one<8, 10, 9, true, false, true, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
    if (d.skew != 0) return launch_one<8, 10, 9, true, true, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
    return launch_one<8, 10, 9, true, false, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
}

}  // namespace sd
