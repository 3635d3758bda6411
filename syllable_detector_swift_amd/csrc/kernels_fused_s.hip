// kernels_fused_s.hip -- the fused engine on the window's symmetry: half the DFT's matrix work, two waves a SIMD, no barriers.
//
// Same path as kernels_fused.hip / kernels_fused_r.hip (reference, root relative:
//   extractPower          Common/CircularShortTimeFourierTransform.swift:280-337
//   processFourierData    Common/SyllableDetector.swift:134-151
//   processNewValue       Common/SyllableDetector.swift:153-217
//   NeuralNet.apply       Common/NeuralNet.swift:294-326, :366-377
//   lastDetected          Common/SyllableDetector.swift:27-31),
// built on what round 3 measured about the part (profiles/r03_energy_probe.txt): under these kernels the chip sits on its
// power limit, so a launch takes what its energy takes, and a v_mfma_f32_16x16x32_f16 costs 8.3 nJ against 1.2 nJ for a
// vector instruction of a wave.  The DFT's 96 matrix instructions per 16 frames were two thirds of the kernel's energy.
//
//   * Every window the reference offers (WindowType.createWindow, CircularShortTimeFourierTransform.swift:19-28: periodic
//     Hamming, Hann, Blackman, rectangular) satisfies w[n] = w[W - n].  Only |X[k]| leaves the transform (:329-333), so the
//     phase reference may sit at the window's centre c = W/2: with m = n - c,
//         Re' X[k] =  sum_{m=0}^{W/2-1} w[c+m] cos(2 pi k m / N) s[m]  +  w[0] cos(pi k W / N) x[0]
//         Im' X[k] = -sum_{m=1}^{W/2-1} w[c+m] sin(2 pi k m / N) d[m]  +  w[0] sin(pi k W / N) x[0]
//     with s[m] = x[c+m] + x[c-m], d[m] = x[c+m] - x[c-m] (s[0] = 2 x[c] against half the coefficient; the frame's first
//     sample x[0] has no partner: its imaginary part rides in the d-GEMM's free slot 0, its real part is 8 multiply-adds).
//     Two GEMMs of K = W/2 instead of one of K = W: 48 matrix instructions per 16 frames instead of 96, a basis of 128
//     registers instead of 256.  The price is vector work: a frame's folded samples are its own (overlapping frames share
//     samples, not sums), so every frame is folded and split into f16 hi + lo by the lanes that own it: ~14 vector
//     instructions a frame instead of ~6.
//   * A basis of 128 registers leaves room for TWO waves per SIMD (256 registers each), which doubles the rate at which
//     the vector instructions issue (tools/ubench/valu_rates: 2.2 clocks against 4.45 for a lone wave) and lets one wave's
//     matrix instructions run under the other's vector work without any hand-placed interleaving.
//   * A wave is a stream processor of its own: it walks a contiguous run of 16-frame tiles of one channel, its samples arrive
//     in a wave-private LDS ring by LDS-DMA (buffer_load ... lds, non-temporal: 7 % less energy per byte and 13 % more bytes per
//     second than plain loads), its tap products live in wave-private LDS rows.  Nothing crosses waves: no workgroup
//     barrier anywhere.
//   * Every FRAME is scaled by its own power of two before the split (its loudest sample goes to [2^13, 2^14)): a result
//     depends on the frame's samples alone, not on what shares a tile or a pass with it.  A click over a quiet cage no longer
//     drags its neighbours to the f16 floor (bench.py's `clicks` record: 500 000 work items on the pass-scaled kernels), and
//     the same samples give the same bits however they are tiled -- streaming and batch results are identical.
//
// Lane (n, g) of a wave: column n of every B operand and result (a frame; slots are permuted so that even frames sit in the
// lanes the LDS serves together with the odd frames of the other lane group: conflict-free ds_read_b128 at hops = 4 mod 8),
// k block 8 g .. 8 g + 7 of an operand, rows 4 g .. 4 g + 3 of a result.
//
// gfx950 only.  wave = 64.

#include "fused_common.hpp"

namespace sd {

namespace {

using namespace fused_dev;

constexpr int kTile = kFusedSTileFrames;       // 16 frames per wave and tile

typedef __attribute__((address_space(3))) void lds_void;

// a * 1 - (the half `hi` or `lo` of h) in fp32: the remainder of a value behind its f16 rounding, exact
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
// two values -> f16 hi pair and f16 lo pair (hi + lo == the value to 2^-22 relative)
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo)
{
    hi = cvt_pk(a, b);
    lo = cvt_pk(rem_lo(a, hi), rem_hi(b, hi));
}

// K2: k-steps of 32 folded positions (W = 64 K2).  GEN: the network class as run-time facts (any transfer functions, with or
// without l2normalize, up to four outputs, log / dB columns); without it the reference's example class (l2normalize, TanSig, one
// linear output).  HQ: quads of hidden units (H <= 4 HQ).  One quad: 8 waves a workgroup, two per SIMD, 256 registers each.
// Wider hidden layers (5 .. 16 units) multiply the first layer's rows -- 9 HQ tap MFMAs a tile, 24 HQ fragment registers, rows
// of 4 HQ T products -- and take 4 waves a workgroup, one per SIMD, with twice the registers and twice the LDS each.
// PADP: 0, or the power of two that divides the hop when the hop is a multiple of 64 floats: the frames of a tile would then
// all start on the same LDS banks, so the ring is laid out with one quad of padding after every PADP floats -- a frame's
// start moves on by one bank group per frame, and inside a frame the padding is a compile-time offset per access.
// PADP == 1 (CS8; hop 128, twice folded; round 6): no padding INSIDE a chunk -- whole chunks are staggered over the banks instead.
// Chunk q sits in slot q mod RC of 1280 bytes at offset 16 S[q & 7], S = {0, 1, 4, 5, 8, 9, 12, 13}: one full-wave DMA
// instruction a chunk (the padded pieces take one each: two a chunk at hop 128, and an LDS-DMA instruction is the dearest
// instruction of the tile), no mirror chunk.  At hop 128 an even frame IS a chunk, an odd frame the second half of one and the first
// half of the next: a lane reads its frame's positions 0 .. 127 through one base and 128 .. 255 through another, both advanced by
// eight slots a tile (q & 7 -- the pad -- is a constant of the lane).  The eight even frames a ds_read_b128 lane group reads with
// lane group g start on quads S[j] (+ 2 g), the eight odd ones with g + 1 on S[j] + 2 (+ 2 g): sixteen different quads of the
// 256-byte bank row -- conflict-free, as the padded ring is.
// F2: the second fold (W == N == 256: K2 == 4, one quad of units, plain ring).  The once-folded positions pair up again (m with
// 128 - m), even and odd bins become GEMMs of their own with K = 64 and one row tile each -- 24 matrix instructions per 16 frames
// instead of 48, a basis of 64 registers -- and the window is applied by the lanes in fp32 (fused_plan.cpp, s2_ok).  A lane then
// folds exactly 64 + 6 samples of its frame, so it reads them from the ring ONCE: the frame's loudest sample is taken from the
// registers the fold reads (16 ds_read_b128 a tile instead of 32).
// NT: row tiles per parity of the twice-folded form: 1 (bands of up to 32 bins), or 2 -- bands of up to 64 bins (11 kHz at 44.1 kHz under
// 256-point frames) stay one launch: 48 + 18 matrix instructions a tile, a basis of 128 registers beside the 70 samples a lane
// holds, so 4 waves a workgroup, one per SIMD, with 512 registers and twice the LDS each.
// SPECT: the |X| (or |X|^2) columns themselves leave, [C][J][F] fp32 (E counts frames then: the plan's stand-in network has
// timeRange 1), nothing of the network runs -- the front half of the wide engine and of syldet_spectrogram*.
template <int K2, bool GEN, int HQ, int NW, int PADP = 0, bool F2 = false, int NT = 1, bool SPECT = false>
__global__ void __launch_bounds__(64 * NW, 1)
fused_s_kernel(const FusedDesc d, const float *__restrict__ samples, int64_t stride, int64_t s_eff, int64_t E,
               float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    constexpr int kWaves = NW;
    const int c = blockIdx.y;
    const int T = d.T, H = d.H, hop = d.hop;
    constexpr int W = 64 * K2;
    const int n_out = GEN ? d.n_out : 1;
    // this wave's segment of the channel
    const int64_t e_b = ((int64_t)blockIdx.x * kWaves + wave) * d.s_seg_evals;
    if (e_b >= E) return;
    const int64_t e_e = (e_b + d.s_seg_evals < E) ? e_b + d.s_seg_evals : E;
    const int seg_len = (int)(e_e - e_b);
    const int tiles = (seg_len + (T - 1) + kTile - 1) / kTile;
    const unsigned e_b32 = (unsigned)e_b;

    // frame of slot n (see the header): slots {0-3, 12-15} take the even frames, {4-11} the odd ones
    const int fr = d.s_perm ? (n < 4 ? 2 * n : (n >= 12 ? 2 * (n - 8) : 2 * (n - 4) + 1)) : n;

    // ---- wave-private LDS: sample ring (RC chunks of 256 floats + one mirror chunk), tap products, a zero quad
    unsigned char *wbase = smem + (size_t)wave * d.s_lds_wave;
    const int RC = PADP == 1 ? d.s_cs8_rc : d.s_ring_chunks, R = RC * 256;
    constexpr bool CS8 = PADP == 1;                   // whole chunks staggered over the banks (hop 128: see the header)
    static_assert(!CS8 || (F2 && HQ == 1 && NT == 1), "staggered chunks: the twice-folded form at hop 128");
    constexpr int kSub = PADP > 1 ? 256 / PADP : 1;   // padded pieces of a chunk of 256 floats
    constexpr int kChunkB = CS8 ? 1280 : 1024 + (PADP > 1 ? 16 * kSub : 0);
    auto sk = [](int i) { return PADP > 1 ? 4 * (i / (PADP > 1 ? PADP : 1)) : 0; };   // padding (floats) in front of position i of a frame / of the ring
    float *ring = reinterpret_cast<float *>(wbase);
    const int PS = d.s_pstride, TP = d.s_tp;          // floats per frame row: 4 TP tap products, sum of squares, floor weight, padding
    float *rows = reinterpret_cast<float *>(wbase + (size_t)(RC + (CS8 ? 0 : 1)) * kChunkB);    // [T - 1 + 16][PS]
    float *zquad = rows + (T - 1 + kTile) * PS;       // 8 HQ floats: HQ zero quads (what taps past timeRange read), HQ quads for stores that have no place
    // Two quads of units at two waves a SIMD: 256 registers hold the basis, 48 registers of first-layer fragments and the
    // loop's working set only if the constants that depend on the lane group alone wait in LDS (4 groups x 20 floats behind the
    // zero quad) and are fetched where they are used -- otherwise 28 registers go to scratch.
    constexpr bool kTbl = HQ == 2 && NW == 8;
    float *gtab = zquad + 8 * HQ + 32 * g;
    {
        const int nz = ((T - 1 + kTile) * PS + 8 * HQ + 128) / 4;          // (+ the table of lane-group constants, where there is one)
        for (int i = lane; i < nz; i += 64) reinterpret_cast<floatx4 *>(rows)[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- constants in registers: the folded basis (A operands; s: real rows against the sums, d: imaginary rows against
    // the differences), the first layer with all taps as rows, the lone sample's real coefficients
    constexpr int KB = F2 ? 1 : K2;                   // (the twice-folded basis lives in b2 below)
    half8 as_[KB][2][2], ad_[KB][2][2];               // [k-step][row tile: bins 0-15, 16-31][hi, lo]
    if (!F2) {
#pragma unroll
        for (int ks = 0; ks < KB; ks++)
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    as_[ks][m][p] = as_half8(reinterpret_cast<const uint32x4 *>(d.sfrag)[(((ks * 2 + 0) * 2 + m) * 2 + p) * 64 + lane]);
                    ad_[ks][m][p] = as_half8(reinterpret_cast<const uint32x4 *>(d.sfrag)[(((ks * 2 + 1) * 2 + m) * 2 + p) * 64 + lane]);
                }
    }
    // twice folded: [k-step][Re even bins, Re odd, Im even, Im odd][hi, lo]; the window coefficients w[128 + m], w[m] of the
    // positions m = 32 ks + 8 g + i this lane folds; cos(pi k / 2) 2^13 for its four even bins; w[192]
    half8 b2[F2 ? 2 : 1][F2 ? 4 * NT : 1][2];        // (row tile tau of GEMM gm: index gm + 4 tau)
    float w1c[F2 ? 16 : 1], w2c[F2 ? 16 : 1], ce[4 * NT], w192 = 0.0f;
#pragma unroll
    for (int i = 0; i < 4 * NT; i++) ce[i] = 0.0f;
    if (F2) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int gm = 0; gm < 4; gm++)
#pragma unroll
                for (int tau = 0; tau < NT; tau++)
#pragma unroll
                    for (int p = 0; p < 2; p++)
                        b2[ks][gm + 4 * tau][p] = as_half8(reinterpret_cast<const uint32x4 *>(d.sfrag2)[(((ks * 4 + gm) * NT + tau) * 2 + p) * 64 + lane]);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            w1c[i] = d.swin2[(g * 16 + i) * 2];
            w2c[i] = d.swin2[(g * 16 + i) * 2 + 1];
        }
#pragma unroll
        for (int i = 0; i < 4 * NT; i++) ce[i] = d.s2c[lane * 16 + i];       // (tile tau's four even rows: 4 tau + i)
        w192 = d.s2c[lane * 16 + 8];
    }
    // (row tile 3 q + m... in table order [m][q]: tap 4 m + g, units 4 q .. 4 q + 3 for lane group g of its result)
    half8 aft[3][HQ][NT][2];                          // (twice folded, 64 bins: one k-step of the tap GEMM per 32 bins)
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int q = 0; q < HQ; q++)
#pragma unroll
            for (int tau = 0; tau < NT; tau++)
#pragma unroll
                for (int p = 0; p < 2; p++)
                    aft[m][q][tau][p] = as_half8(reinterpret_cast<const uint32x4 *>(F2 ? (HQ == 1 ? d.afrag_t2 : d.afrag_w2) : (HQ == 1 ? d.afrag_t : d.afrag_w))[(((m * HQ + q) * NT + tau) * 2 + p) * 64 + lane]);
    float cre[8];                                     // w[0] cos(pi k W / N) 2^13 for this lane's bins 4 g + i, 16 + 4 g + i
#pragma unroll
    for (int i = 0; i < 8; i++) cre[i] = F2 ? 0.0f : d.slone[lane * 8 + i];
    if (kTbl && n == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) gtab[i] = cre[i];
    }

    const float *row = samples + (int64_t)c * stride;
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(row), 0, (int)(s_eff * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(
        outputs ? outputs + (int64_t)c * E * n_out : nullptr, 0, outputs ? (int)(E * n_out * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t flg_rs = __builtin_amdgcn_make_buffer_rsrc(flags ? flags + (int64_t)c * E : nullptr, 0, flags ? (int)E : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t spc_rs = __builtin_amdgcn_make_buffer_rsrc(
        SPECT ? d.spect_out + (int64_t)c * E * d.F : nullptr, 0, SPECT ? (int)(E * d.F * 4) : 0, 0x00020000);

    // ---- the sample stream: chunk q holds samples [256 q, 256 q + 256) behind the segment's first one, in ring slot q mod RC;
    // slot 0's chunks are written a second time behind the ring (the mirror), so that a frame's reads never wrap.
    // A chunk may be issued once the chunk it replaces is dead: while tile t is read, chunks up to RC - 1 + floor(16 hop t / 256).
    const unsigned org = (unsigned)((e_b * hop + d.gap) * 4);            // byte offset of the segment's first sample in its row (row < 4 GB: launcher)
    int cn = 0, slot = 0;                                                // next chunk, its ring slot
    auto issue_upto = [&](int last) {
        while (cn <= last) {
            const unsigned voff = org + (unsigned)cn * 1024u + (unsigned)lane * 16u;
            if (CS8) {
                const int pad = ((cn & 6) << 1) | (cn & 1);              // S[cn & 7]
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void *)(wbase + slot * kChunkB + 16 * pad), 16, voff, 0, 0, 2 /* nt */);
            } else if (PADP == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void *)(wbase + slot * 1024), 16, voff, 0, 0, 2 /* nt */);
                if (slot == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void *)(wbase + RC * 1024), 16, voff, 0, 0, 2);
            } else {
                // (the DMA writes base + 16 lane: one instruction per padded piece, its lanes only, the base moved on by the padding)
#pragma unroll
                for (int k = 0; k < kSub; k++)
                    if (lane / (64 / kSub) == k) {
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void *)(wbase + slot * kChunkB + 16 * k), 16, voff, 0, 0, 2);
                        if (slot == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rs, (lds_void *)(wbase + RC * kChunkB + 16 * k), 16, voff, 0, 0, 2);
                    }
            }
            cn++;
            slot = slot + 1 == RC ? 0 : slot + 1;
        }
    };
    const int span = (kTile - 1) * hop + W;                              // samples under one tile
    auto need = [&](int t) { return (kTile * hop * t + span - 1) >> 8; };
    auto allowed = [&](int t) { return RC - 1 + ((kTile * hop * t) >> 8); };

    // ---- per-lane LDS places.  fo: this lane's frame inside the ring (floats), advanced by 16 hop a tile.
    unsigned fo = (unsigned)(hop * fr);                                  // < R (launcher: 16 hop <= R)
    // CS8: byte offsets of the slots that hold this lane's positions 0 .. 127 (sb1) and 128 .. 255 (sb2), and what does not change:
    // the chunks' pads, and which half of its chunk each part is
    unsigned sb1 = (unsigned)((fr >> 1) * kChunkB), sb2 = (unsigned)(((fr >> 1) + (fr & 1)) * kChunkB);
    const int jq1 = (fr >> 1) & 7, jq2 = ((fr >> 1) + (fr & 1)) & 7;
    const unsigned cb1 = (unsigned)(16 * (((jq1 & 6) << 1) | (jq1 & 1)) + ((fr & 1) ? 512 : 0));
    const unsigned cb2 = (unsigned)(16 * (((jq2 & 6) << 1) | (jq2 & 1)) + ((fr & 1) ? 0 : 512));
    float *prow = rows + (T - 1 + fr) * PS;                              // this frame's row of tap products
    const float *erow = rows + n * PS;                                   // evaluation n: rows n .. n + T - 1
    const float *pv_p[3], *sv_p[3];
#pragma unroll
    for (int tt = 0; tt < 3; tt++) {
        const int t = g + 4 * tt;
        // (taps past timeRange: their products are 0 * column, which is NaN for a column with a NaN in it -- read zeros)
        pv_p[tt] = t < T ? erow + t * PS + 4 * HQ * t : zquad;
        sv_p[tt] = t < T ? erow + t * PS + 4 * HQ * TP : zquad;
    }
    // tile m of the tap products holds tap 4 m + g of this frame: taps past TP have no place in the row
    float *pt_p[3];
#pragma unroll
    for (int m = 0; m < 3; m++) pt_p[m] = 4 * m + g < TP ? prow + 4 * HQ * (4 * m + g) : zquad + 4 * HQ;

    const float c_b1 = GEN ? (g < n_out ? d.b1[g] : 0.0f) : d.b1[0];
    float lean_oa = 0.0f, lean_og = 1.0f, lean_ob = 0.0f;
    if (d.n_out_fns == 1) {
        const int o = (GEN && g < n_out) ? g : 0;
        lean_oa = d.out_params[0]; lean_og = d.out_params[1 + o]; lean_ob = d.out_params[1 + n_out + o];
    }
    // this lane group's hidden units: 4 q + g
    float b0g[HQ], w1g[HQ], w1o[4][HQ], rvg[HQ];      // (rvg: (W0 o a) . 1, what the normalisers' offset meets)
#pragma unroll
    for (int q = 0; q < HQ; q++) {
        const int u = 4 * q + g;
        rvg[q] = (GEN && u < H) ? d.rvec[u] : 0.0f;
        b0g[q] = u < H ? d.bias0[u] : 0.0f;
        w1g[q] = u < H ? d.w1[u] : 0.0f;
#pragma unroll
        for (int o = 0; o < 4; o++) w1o[o][q] = (GEN && u < H && o < n_out) ? d.w1[o * H + u] : 0.0f;
    }
    if (kTbl && n == 0) {
#pragma unroll
        for (int q = 0; q < HQ; q++) {
            gtab[8 + q] = b0g[q];
            gtab[8 + HQ + q] = w1g[q];
            gtab[8 + 6 * HQ + q] = rvg[q];
#pragma unroll
            for (int o = 0; o < 4; o++) gtab[8 + 2 * HQ + o * HQ + q] = w1o[o][q];
        }
    }
    const bool multi = GEN && n_out > 1;
    const double thr_g = d.thresholds[(GEN && g < n_out) ? g : 0];
    if (kTbl && n == 0) {                              // (the output stage's constants wait in the table too)
        gtab[22] = c_b1; gtab[23] = lean_oa; gtab[24] = lean_og; gtab[25] = lean_ob;
        *reinterpret_cast<double *>(gtab + 26) = thr_g;
    }
    const bool counts = !GEN || n_out == 1 ? true : (g < n_out && (g == 0 || d.rule == 1));
    const int tf0 = GEN ? d.tf0 : 0, tf1 = GEN ? d.tf1 : 2, norm = GEN ? d.norm : 1;
    const int scaling = GEN ? d.scaling : 0;          // linear |X|, or ln / 20 log10 of it in front of the chain (SyllableDetector.swift:184-212)
    const float klog = scaling == 1 ? 0.6931471805599453f : 6.020599913279624f;      // ln 2, 20 log10 2
    bool binv[8 * NT];                                // which of this lane's 8 (16) bins are band bins (rows past F are zeros: log 0)
#pragma unroll
    for (int i = 0; i < 8 * NT; i++) {
        const int tau = i >> 3, ii = i & 7;
        binv[i] = (F2 ? 32 * tau + (ii < 4 ? 8 * g + d.s2_pe + 2 * ii : 8 * g + d.s2_po + 2 * (ii - 4)) : (ii < 4 ? 4 * g + ii : 16 + 4 * g + (ii - 4))) < d.F;
    }
    const float kmag = pow2f(-13 - d.col_shift);
    const bool guard_on = d.fix.counters != nullptr;
    const float guard_k = norm == 1 ? d.guard_r : (norm == 0 ? d.guard_rel_r : d.guard_range_r);

    // ---- prologue: tile 0's samples
    issue_upto(need(0) < allowed(0) ? need(0) : allowed(0));
    int se_ref = 0;                                   // products are stored relative to the segment's first frame that has a level

#ifdef SYLDET_S_STAMPS                // (diagnostic build: shader clocks a wave spends waiting for its samples' DMA / for the LDS, and in all)
    unsigned long long st_vm = 0, st_lg = 0;
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#define SD_STAMP(var, stmt) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); stmt; \
                              const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); var += t1_ - t0_; }
#else
#define SD_STAMP(var, stmt) { stmt; }
#endif
    // Everything the prologue fetched -- basis, tables, tile 0's chunks -- is here before the loop, and the COMPILER is told so
    // (the builtin, not an asm string): otherwise its wait-count model carries the prologue's loads into the loop as still
    // pending and guards their first uses in every iteration with vmcnt(6) / (1) / (0) -- waits that in steady state only make a
    // wave wait for the NEXT tile's DMA chunks a quarter, a half and four fifths of the way through the current tile.
#ifndef SYLDET_S_NO_PREWAIT            // (diagnostic build: tools/variant_libs.sh kernels_fused_s.hip noprewait -DSYLDET_S_NO_PREWAIT)
    __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
    // (round 6 experiment, MEASUREMENTS R6.5: a static priority for one wave of every SIMD's pair -- MI355X_MICROARCH.md, "Two waves per
    // SIMD", item 4 -- built only under these defines)
#if defined(SYLDET_S_SETPRIO_SECOND)
    if (wave >= kWaves / 2) __builtin_amdgcn_s_setprio(1);
#elif defined(SYLDET_S_SETPRIO_FIRST)
    if (wave < kWaves / 2) __builtin_amdgcn_s_setprio(1);
#endif
    for (int t = 0; t < tiles; t++) {
        // tile t's chunks have landed (behind them in the queue: nothing but the two result stores of tile t - 1)
        // (SPECT: the five column stores of tile t - 1)
        SD_STAMP(st_vm, if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else if (SPECT) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"));
        const float *f1 = reinterpret_cast<const float *>(wbase + sb1 + cb1);       // (CS8) positions 0 .. 127 of this lane's frame
        const float *f2 = reinterpret_cast<const float *>(wbase + sb2 + cb2);       //       positions 128 .. 255, at f2[position - 128]
        const float *fp = CS8 ? f1 : ring + fo + sk((int)fo);    // this lane's frame: W samples from here (the mirror makes them contiguous)

        // ---- twice folded: this lane's 64 + 6 samples of its frame, read once.  For m0 = 32 ks + 8 g:
        //   P1 = x[m0 .. m0+7], P3 = x[128+m0 .. +7] (two quads each), x[128-m0-i] and x[256-m0-i], i = 0..7: two quads
        //   [120-m0, 128-m0), [248-m0, 256-m0) and the words x[128-m0], x[256-m0]; x[64] and x[192] for the self-paired position
        floatx4 P1[2][2], P3[2][2], P2[2][2], P4[2][2];
        float P2w[2], P4w[2], x64 = 0.0f, x192 = 0.0f;
        if (F2) {
#pragma unroll
            // (padded ring: position i of the frame sits at fp[i + sk(i)]; the quads lie inside one piece each, so the padding
            // is a constant per access -- only the word x[128 - m0] of lane group 0, k-step 0 is on the far side of one)
            for (int ks = 0; ks < 2; ks++) {
                if (CS8) {
                    const float *a1 = f1 + 32 * ks + 8 * g, *a2 = f2 + 32 * ks + 8 * g;
                    const float *b1 = f1 + 120 - 32 * ks - 8 * g, *b3 = f2 + 120 - 32 * ks - 8 * g;
                    P1[ks][0] = *reinterpret_cast<const floatx4 *>(a1);
                    P1[ks][1] = *reinterpret_cast<const floatx4 *>(a1 + 4);
                    P3[ks][0] = *reinterpret_cast<const floatx4 *>(a2);
                    P3[ks][1] = *reinterpret_cast<const floatx4 *>(a2 + 4);
                    P2[ks][0] = *reinterpret_cast<const floatx4 *>(b1);
                    P2[ks][1] = *reinterpret_cast<const floatx4 *>(b1 + 4);
                    P2w[ks] = (ks == 0) ? (g == 0 ? f2 : b1 + 8)[0] : b1[8];       // (position 128 - m0: the second part's first word for lane group 0)
                    P4[ks][0] = *reinterpret_cast<const floatx4 *>(b3);
                    P4[ks][1] = *reinterpret_cast<const floatx4 *>(b3 + 4);
                    P4w[ks] = b3[8];                      // (g == 0, ks == 0: position 256, the next frame's -- read, never used)
                    continue;
                }
                const float *a = fp + 32 * ks + 8 * g, *b = fp + 120 - 32 * ks - 8 * g;
                P1[ks][0] = *reinterpret_cast<const floatx4 *>(a);
                P1[ks][1] = *reinterpret_cast<const floatx4 *>(a + 4);
                P3[ks][0] = *reinterpret_cast<const floatx4 *>(a + 128 + sk(128));
                P3[ks][1] = *reinterpret_cast<const floatx4 *>(a + 132 + sk(128));
                P2[ks][0] = *reinterpret_cast<const floatx4 *>(b + sk(64));
                P2[ks][1] = *reinterpret_cast<const floatx4 *>(b + 4 + sk(64));
                P2w[ks] = (PADP && ks == 0) ? b[8 + (g == 0 ? sk(128) : sk(64))] : b[8 + sk(64)];
                P4[ks][0] = *reinterpret_cast<const floatx4 *>(b + 128 + sk(192));
                P4[ks][1] = *reinterpret_cast<const floatx4 *>(b + 132 + sk(192));
                P4w[ks] = b[136 + sk(192)];             // (g == 0, ks == 0: x[256], the next frame's -- read, never used)
            }
            x64 = CS8 ? f1[64] : fp[64 + sk(64)];
            x192 = CS8 ? f2[64] : fp[192 + sk(192)];
            // the raw samples of this tile are dead as soon as they are in registers: all of the next tile's chunks, a whole tile
            // of arithmetic ahead of their use
            SD_STAMP(st_lg, asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"));
            if (t + 1 < tiles) issue_upto(need(t + 1));
        }
        // ---- the frame's own scale from its loudest sample (this lane looks at a quarter of the frame)
        float amax;
#ifdef SYLDET_S_NOMAX                // (diagnostic knock-outs, tools/s_knockouts.sh: wrong results by construction, never the shipped library)
        amax = 1.0f;
#else
        {
            float m0 = 0.0f, m1 = 0.0f;
            if (F2) {
                // (every word read above lies inside the frame except x[256]: the four lanes' sets cover the frame)
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        m0 = absmax3(m0, P1[ks][h][0], P1[ks][h][1]);
                        m1 = absmax3(m1, P1[ks][h][2], P1[ks][h][3]);
                        m0 = absmax3(m0, P3[ks][h][0], P3[ks][h][1]);
                        m1 = absmax3(m1, P3[ks][h][2], P3[ks][h][3]);
                        m0 = absmax3(m0, P2[ks][h][0], P2[ks][h][1]);
                        m1 = absmax3(m1, P2[ks][h][2], P2[ks][h][3]);
                        m0 = absmax3(m0, P4[ks][h][0], P4[ks][h][1]);
                        m1 = absmax3(m1, P4[ks][h][2], P4[ks][h][3]);
                    }
                m0 = absmax3(m0, P2w[0], P2w[1]);
                m1 = absmax3(m1, g == 0 ? 0.0f : P4w[0], P4w[1]);
                m0 = absmax3(m0, x64, x192);
            } else {
                const floatx4 *q0 = reinterpret_cast<const floatx4 *>(fp + (W / 4) * g + sk((W / 4) * g));   // (a quarter never crosses a padding)
#pragma unroll
                for (int q = 0; q < W / 16; q++) {
                    const floatx4 v = q0[q];
                    m0 = absmax3(m0, v[0], v[1]);
                    m1 = absmax3(m1, v[2], v[3]);
                }
            }
            amax = fmaxf(m0, m1);                      // (v_max3 drops NaNs: plain non-negative numbers from here on)
            auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(amax), __float_as_uint(amax), false, false);
            amax = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
            r = __builtin_amdgcn_permlane32_swap(__float_as_uint(amax), __float_as_uint(amax), false, false);
            amax = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
        }
#endif
        // status of the frame for the precision guard: 0 fine, 1 silent (its column is exact zeros), 2 the grid cannot hold it
        // (an infinite sample, or a level above 2^113)
        const int ex = (int)((__float_as_uint(amax) >> 23) & 0xffu);
        // 2^se puts the loudest sample into [2^13, 2^14); twice folded into [2^12, 2^13): a folded position is the sum of four
        // samples there, under window coefficients that add up to 2 at most (rectangular)
        int se = (F2 ? 139 : 140) - ex;
        const int fst = amax > 0.0f ? ((ex == 255 || se < -100) ? 2 : 0) : 1;
        se = amax > 0.0f ? (se < -100 ? -100 : (se > 113 ? 113 : se)) : 0;
        const float sx = pow2f(se);
        if (t == 0) {                                 // the segment's reference exponent: the loudest frame of its first tile
            const float tm = wave_max_nonneg(amax);
            const int exr = (int)((__float_as_uint(tm) >> 23) & 0xffu);
            int r0 = (F2 ? 139 : 140) - exr;
            r0 = tm > 0.0f ? (r0 < -100 ? -100 : (r0 > 113 ? 113 : r0)) : 0;
            se_ref = __builtin_amdgcn_readfirstlane(r0);
        }

        // the next tile's samples, as far as the ring has room while this tile is read
        if (!F2 && t + 1 < tiles) issue_upto(need(t + 1) < allowed(t) ? need(t + 1) : allowed(t));

        // ---- the folded DFT: per k-step this lane folds, scales and splits 8 positions of its frame, then 12 MFMAs
        const float xl = fp[0] * sx;                  // the frame's first sample (no partner)
        floatx4 acc[4 * NT];                          // re 0-15, re 16-31, im 0-15, im 16-31 (twice folded: re even, re odd, im even, im odd of row tile tau at 4 tau + ..)
#pragma unroll
        for (int i = 0; i < 4 * NT; i++) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        float a64 = 0.0f;                             // twice folded: re even, re odd, im even, im odd; the self-paired position's sum
        const float *xpb = fp + W / 2 + 8 * g, *xmb = fp + W / 2 - 8 * g;
        const float *xmz = xmb + ((PADP && g == 0) ? 4 : 0);     // (the word c - m0 of lane group 0 sits on a piece's first position)
#ifndef SYLDET_S_NODFT
        if (F2) {
            a64 = w192 * fmaf(x192, sx, x64 * sx);
            const float b64 = w192 * fmaf(x192, sx, -(x64 * sx));
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                // position m = 32 ks + 8 g + i:  xa = x[128 + m], xb = x[128 - m], xc = x[256 - m], xd = x[m]
                const float xa[8] = {P3[ks][0][0], P3[ks][0][1], P3[ks][0][2], P3[ks][0][3], P3[ks][1][0], P3[ks][1][1], P3[ks][1][2], P3[ks][1][3]};
                const float xd[8] = {P1[ks][0][0], P1[ks][0][1], P1[ks][0][2], P1[ks][0][3], P1[ks][1][0], P1[ks][1][1], P1[ks][1][2], P1[ks][1][3]};
                const float xb[8] = {P2w[ks], P2[ks][1][3], P2[ks][1][2], P2[ks][1][1], P2[ks][1][0], P2[ks][0][3], P2[ks][0][2], P2[ks][0][1]};
                const float xc[8] = {P4w[ks], P4[ks][1][3], P4[ks][1][2], P4[ks][1][1], P4[ks][1][0], P4[ks][0][3], P4[ks][0][2], P4[ks][0][1]};
                float ap[8], am[8], bm[8], bp[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float t1 = xb[i] * sx, t2 = xd[i] * sx;
                    float s1 = fmaf(xa[i], sx, t1), d1 = fmaf(xa[i], sx, -t1);       // once folded about the centre: position m
                    float s2 = fmaf(xc[i], sx, t2), d2 = fmaf(xc[i], sx, -t2);       //                               position 128 - m
                    if (ks == 0 && i == 0) {
                        // m = 0 has no partners: the centre sample x[128] under w[128] and the frame's first sample x[0] under w[0]
                        s1 = g == 0 ? xa[0] * sx : s1;
                        s2 = g == 0 ? t2 : s2;
                    }
                    const float w1 = w1c[ks * 8 + i], w2 = w2c[ks * 8 + i];
                    const float p = w2 * s2, q = w2 * d2;
                    ap[i] = fmaf(w1, s1, p);                                         // even bins, real
                    am[i] = fmaf(w1, s1, -p);                                        // odd bins, real
                    bm[i] = fmaf(w1, d1, -q);                                        // even bins, imaginary
                    bp[i] = fmaf(w1, d1, q);                                         // odd bins, imaginary
                }
                if (ks == 0) {                        // slot 0 of the imaginary rows: sin 0 -- free; the odd bins' takes the self-paired position
                    bm[0] = g == 0 ? 0.0f : bm[0];
                    bp[0] = g == 0 ? b64 : bp[0];
                }
                uint32x4 vh[4], vl[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    unsigned h, l;
                    split2(ap[2 * j], ap[2 * j + 1], h, l);
                    vh[0][j] = h; vl[0][j] = l;
                    split2(am[2 * j], am[2 * j + 1], h, l);
                    vh[1][j] = h; vl[1][j] = l;
                    split2(bm[2 * j], bm[2 * j + 1], h, l);
                    vh[2][j] = h; vl[2][j] = l;
                    split2(bp[2 * j], bp[2 * j + 1], h, l);
                    vh[3][j] = h; vl[3][j] = l;
                }
#pragma unroll
                for (int gm = 0; gm < 4 * NT; gm++) acc[gm] = mfma(b2[ks][gm][0], as_half8(vh[gm & 3]), acc[gm]);
#pragma unroll
                for (int gm = 0; gm < 4 * NT; gm++) acc[gm] = mfma(b2[ks][gm][0], as_half8(vl[gm & 3]), acc[gm]);
#pragma unroll
                for (int gm = 0; gm < 4 * NT; gm++) acc[gm] = mfma(b2[ks][gm][1], as_half8(vh[gm & 3]), acc[gm]);
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < KB; ks++) {
            // x[c + m0 + i], i = 0..7, and x[c - m0 - i]: words c-m0-8 .. c-m0-1 as two quads, and the word c - m0
            const int skp = sk(W / 2 + 32 * ks), skm = sk(W / 2 - 32 * ks - 32), skw = sk(W / 2 - 32 * ks);   // (constants once unrolled)
            const floatx4 p0 = *reinterpret_cast<const floatx4 *>(xpb + 32 * ks + skp), p1 = *reinterpret_cast<const floatx4 *>(xpb + 32 * ks + 4 + skp);
            const floatx4 q1 = *reinterpret_cast<const floatx4 *>(xmb - 32 * ks - 8 + skm), q2 = *reinterpret_cast<const floatx4 *>(xmb - 32 * ks - 4 + skm);
            const float q0 = (skw != skm ? xmz : xmb)[-32 * ks + skm];
            const float xp[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
            const float xm[8] = {q0, q2[3], q2[2], q2[1], q2[0], q1[3], q1[2], q1[1]};
            float s[8], dd[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float tm = xm[i] * sx;
                s[i] = fmaf(xp[i], sx, tm);
                dd[i] = fmaf(xp[i], sx, -tm);
            }
            if (ks == 0) dd[0] = g == 0 ? xl : dd[0];  // slot 0 of the differences carries the lone sample
            uint32x4 sh, sl, dh, dl;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                unsigned h, l;
                split2(s[2 * j], s[2 * j + 1], h, l);
                sh[j] = h; sl[j] = l;
                split2(dd[2 * j], dd[2 * j + 1], h, l);
                dh[j] = h; dl[j] = l;
            }
            const half8 bsh = as_half8(sh), bsl = as_half8(sl), bdh = as_half8(dh), bdl = as_half8(dl);
#pragma unroll
            for (int m = 0; m < 2; m++) acc[m] = mfma(as_[ks][m][0], bsh, acc[m]);
#pragma unroll
            for (int m = 0; m < 2; m++) acc[2 + m] = mfma(ad_[ks][m][0], bdh, acc[2 + m]);
#pragma unroll
            for (int m = 0; m < 2; m++) acc[m] = mfma(as_[ks][m][0], bsl, acc[m]);
#pragma unroll
            for (int m = 0; m < 2; m++) acc[2 + m] = mfma(ad_[ks][m][0], bdl, acc[2 + m]);
#pragma unroll
            for (int m = 0; m < 2; m++) acc[m] = mfma(as_[ks][m][1], bsh, acc[m]);
#pragma unroll
            for (int m = 0; m < 2; m++) acc[2 + m] = mfma(ad_[ks][m][1], bdh, acc[2 + m]);
        }
        }
#else
        acc[0][0] = xpb[0]; acc[1][1] = xmb[0];
#endif
        // the raw samples of this tile are dead (every read of them has returned): the rest of the next tile's chunks
        if (!F2) {
            SD_STAMP(st_lg, asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"));
            if (t + 1 < tiles) issue_upto(need(t + 1));
        }

        // ---- |X| (zvabs / 2, CircularShortTimeFourierTransform.swift:329-333) of this lane's 8 bins, the frame's sum of squares,
        // the f16 hi + lo split of the column under the frame's own column exponent, the tap products of the first layer.
        // acc holds X 2^(se + 13); cval = |X| 2^(se - col_shift).
        float cval[8 * NT], mss = 0.0f;
        float cre_l[8];
        if (kTbl) {                                   // (the address passes through an opaque statement: the fetch stays in the loop)
            const float *tp = gtab;
            asm volatile("" : "+v"(tp));
            const floatx4 c0 = *reinterpret_cast<const floatx4 *>(tp), c1 = *reinterpret_cast<const floatx4 *>(tp + 4);
#pragma unroll
            for (int i = 0; i < 4; i++) { cre_l[i] = c0[i]; cre_l[4 + i] = c1[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) cre_l[i] = cre[i];
        }
#pragma unroll
        for (int i = 0; i < 8 * NT; i++) {
            // once folded: acc = re 0-15, re 16-31, im 0-15, im 16-31, the first sample's real part on top;
            // twice folded: acc = re even, re odd, im even, im odd (of row tile tau), the self-paired position's sum on top of the even real rows
            const int tau = i >> 3, ii = i & 7;
            const float re = F2 ? (ii < 4 ? fmaf(ce[4 * tau + ii], a64, acc[4 * tau][ii]) : acc[4 * tau + 1][ii & 3]) : fmaf(cre_l[ii], xl, acc[ii >> 2][ii & 3]);
            const float im = F2 ? (ii < 4 ? acc[4 * tau + 2][ii] : acc[4 * tau + 3][ii & 3]) : acc[2 + (ii >> 2)][ii & 3];
            cval[i] = __builtin_amdgcn_sqrtf(fmaf(re, re, im * im)) * kmag;
        }
        if constexpr (SPECT) {
            // columns only: |X| = cval 2^(col_shift - se) in true units (zvabs / 2, :329-333), or |X|^2 (zvmags / 4, :270-274), of
            // this lane's band bins -> HBM; this lane's frame is frame fr of the tile (the slot permutation), evaluation = frame (timeRange 1)
            static_assert(NT == 1 && F2, "the spectrogram instantiation is the twice-folded one");
            const int er = kTile * t + fr;
            const bool vld = er < seg_len;
            const int ush = d.col_shift - se;
            const float up = pow2f(ush < -126 ? -126 : (ush > 126 ? 126 : ush));
            const unsigned rowo = (e_b32 + (unsigned)er) * (unsigned)d.F;
            // this lane's eight bins 8 g .. 8 g + 7 in band order (the even / odd GEMMs' rows interleave; which comes first
            // depends on the parity of the band's first bin): two quads a lane, 32 contiguous bytes of its frame's row
            const bool odd_first = d.s2_pe != 0;
            float v8[8];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float me = cval[i] * up, mo = cval[4 + i] * up;
                const float a = odd_first ? mo : me, b = odd_first ? me : mo;
                v8[2 * i] = d.spect_power ? a * a : a;
                v8[2 * i + 1] = d.spect_power ? b * b : b;
            }
            // whole quads as one store each; the band's last, partial quad (F mod 4 bins) as single words
            const int fq = d.F >> 2, rem = d.F & 3;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int Q = 2 * g + q;
                const uint32x4 w = {__float_as_uint(v8[4 * q]), __float_as_uint(v8[4 * q + 1]), __float_as_uint(v8[4 * q + 2]), __float_as_uint(v8[4 * q + 3])};
                __builtin_amdgcn_raw_buffer_store_b128(w, spc_rs, (vld && Q < fq) ? (rowo + 4u * (unsigned)Q) * 4u : 0xFFFFFFFFu, 0, 0);
            }
            {
                const bool mine = vld && (fq >> 1) == g;                 // the partial quad is this lane group's quad fq & 1
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const float pv = (fq & 1) ? v8[4 + j] : v8[j];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pv), spc_rs, (mine && j < rem) ? (rowo + 4u * (unsigned)fq + (unsigned)j) * 4u : 0xFFFFFFFFu, 0, 0);
                }
            }
            // the precision guard: a frame the grid cannot hold (an infinite sample, a level above 2^113) is recomputed from its
            // samples -- every other frame carries its own scale.  16 frames of a wave are one work item.
            if (guard_on && __builtin_amdgcn_ballot_w64(vld && fst == 2) != 0ull) {
                const int lo = kTile * t, hi = lo + kTile < seg_len ? lo + kTile : seg_len;
                if (lane == 0 && hi > lo) {
                    const unsigned sl = atomicAdd(d.fix.counters, 1u);
                    if (sl < d.fix.capacity) d.fix.items[sl] = FixItem{c, e_b32 + (unsigned)lo, hi - lo, 1};
                    else d.fix.counters[3] = 1u;
                }
            }
            fo += (unsigned)(kTile * hop);
            fo = fo >= (unsigned)R ? fo - (unsigned)R : fo;
            if (CS8) {
                sb1 += 8u * kChunkB; sb1 = sb1 >= (unsigned)(RC * kChunkB) ? sb1 - (unsigned)(RC * kChunkB) : sb1;
                sb2 += 8u * kChunkB; sb2 = sb2 >= (unsigned)(RC * kChunkB) ? sb2 - (unsigned)(RC * kChunkB) : sb2;
            }
            continue;
        }
        if (GEN && scaling != 0) {
            // log / dB columns: |X| = cval 2^(col_shift - se) in true units; v_log_f32 is log2; ln 0 = -inf as in the reference
            const float off = (float)(d.col_shift - se) * klog;
#pragma unroll
            for (int i = 0; i < 8 * NT; i++) cval[i] = binv[i] ? fmaf(__builtin_amdgcn_logf(cval[i]), klog, off) : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 8 * NT; i++) mss = fmaf(cval[i], cval[i], mss);
        mss = xor32_sum(xor16_sum(mss));
        // products and sums of squares are stored relative to the segment's reference exponent (* 2^dsc, * 4^dsc): a window
        // straddles frames of different scales.  Frames 2^45 away from it, and frames the grid cannot hold, condemn their windows.
        // (log / dB columns are absolute numbers: no relative scale, and no grid floor to guard beyond what no grid can hold)
        int dsc = (GEN && scaling != 0) ? 0 : se_ref - se;
        const bool far = dsc > 45 || dsc < -45 || fst == 2;
        dsc = dsc < -45 ? -45 : (dsc > 45 ? 45 : dsc);
        // The frame's own column exponent: its column is split at the scale that puts its norm into [2^12, 2^13); with ex the
        // biased exponent of mss, tb = floor((ex + 1) / 2) = floor(log2 sqrt(mss)) + 64, clamped to [16, 80]: fs_up = 2^(76 - tb).
        unsigned tb = ((__float_as_uint(mss) + 0x800000u) >> 1) & 0x7f800000u;
        tb = (unsigned)min(max((int)tb, 16 << 23), 80 << 23);
        const float fs_up = __uint_as_float((203u << 23) - tb);
        const float fs_ring = __uint_as_float(tb + ((unsigned)(dsc + 51) << 23));
        uint32x4 bh[NT], bl[NT];
#pragma unroll
        for (int j = 0; j < 4 * NT; j++) {
            unsigned h, l;
            split2(cval[2 * j] * fs_up, cval[2 * j + 1] * fs_up, h, l);
            bh[j >> 2][j & 3] = h; bl[j >> 2][j & 3] = l;
        }
        floatx4 pt[3][HQ];
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int q = 0; q < HQ; q++) {
                pt[m][q] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tau = 0; tau < NT; tau++) {
                    pt[m][q] = mfma(aft[m][q][tau][0], as_half8(bh[tau]), pt[m][q]);
                    pt[m][q] = mfma(aft[m][q][tau][0], as_half8(bl[tau]), pt[m][q]);
                    pt[m][q] = mfma(aft[m][q][tau][1], as_half8(bh[tau]), pt[m][q]);
                }
            }
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int q = 0; q < HQ; q++) *reinterpret_cast<floatx4 *>(pt_p[m] + 4 * q) = pt[m][q] * fs_ring;
        {
            // the frame's sum of squares relative to the reference, and the weight of its grid floor there: 4^dsc (0 for a
            // silent frame: exact zeros; +inf for a frame the grid cannot hold)
            float st0 = mss * pow2f(2 * dsc), st1 = 0.0f;
            if (GEN && norm == 2) {                   // Normalize (NeuralNet.swift:69-96): the frame's smallest and largest band bin
                float mn = INFINITY, mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < 8 * NT; i++) {
                    mn = binv[i] ? fminf(mn, cval[i]) : mn;
                    mx = binv[i] ? fmaxf(mx, cval[i]) : mx;
                }
                auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(mn), __float_as_uint(mn), false, false);
                mn = fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
                r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mn), __float_as_uint(mn), false, false);
                mn = fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
                r = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
                r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
                st0 = mn * pow2f(dsc);
                st1 = mx * pow2f(dsc);
            } else if (GEN && norm == 3) {            // NormalizeStd (:105-108): the frame's mean and sum of squared deviations
                float sm = 0.0f;
#pragma unroll
                for (int i = 0; i < 8 * NT; i++) sm += binv[i] ? cval[i] : 0.0f;
                sm = xor32_sum(xor16_sum(sm));
                const float mean = sm / (float)d.F;
                float m2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 8 * NT; i++) {
                    const float dl = cval[i] - mean;
                    m2 = binv[i] ? fmaf(dl, dl, m2) : m2;
                }
                m2 = xor32_sum(xor16_sum(m2));
                st0 = mean * pow2f(dsc);
                st1 = m2 * pow2f(2 * dsc);
            }
            if (GEN && norm == 0) st1 = st0;           // no normaliser: the frame's sum of squares once more, for the window's level (the loudness guard)
            const float fw = far ? INFINITY : ((fst == 1 || (GEN && scaling != 0)) ? 0.0f : pow2f(2 * dsc));
            if (g == 0) *reinterpret_cast<floatx4 *>(prow + 4 * HQ * TP) = floatx4{st0, fw, st1, 0.0f};
        }

        // ---- the tile's 16 evaluations: evaluation n ends on frame n of the tile; its taps are rows n .. n + T - 1.  Lane
        // group g takes taps g, g + 4, g + 8 and, after the halving butterfly, hidden unit g.
        floatx4 zp[HQ];
        float ssp, fwp, st1p = 0.0f;                  // this lane group's share of the window's statistics (its taps g, g + 4, g + 8)
        float sq0[3];                                 // (normalizestd: the frames' means once more)
        // The last T - 1 frames' rows go to the front for the next tile.  Their reads are issued HERE, with the evaluation's own
        // (one wait for both; clamped indices instead of masked lanes), their writes behind the evaluation's reads -- a wave's LDS
        // operations execute in order, and source rows 16 .. never overlap destination rows 0 .. T - 2 (T - 1 < 16).
        const int nq = (T - 1) * PS / 4;
        floatx4 cy0 = {0.f, 0.f, 0.f, 0.f}, cy1 = cy0, cy2 = cy0;
        if (HQ == 1) {                                // at most 143 quads (two or three a lane)
            const floatx4 *src = reinterpret_cast<const floatx4 *>(rows + kTile * PS);
            cy0 = src[lane < nq ? lane : 0];
            cy1 = src[lane + 64 < nq ? lane + 64 : 0];
            if (nq > 128) cy2 = src[lane + 128 < nq ? lane + 128 : 0];
        }
        {
            floatx4 pv[3][HQ];
#pragma unroll
            for (int tt = 0; tt < 3; tt++)
#pragma unroll
                for (int q = 0; q < HQ; q++) pv[tt][q] = *reinterpret_cast<const floatx4 *>(pv_p[tt] + 4 * q);
            const floatx4 s0 = *reinterpret_cast<const floatx4 *>(sv_p[0]), s1 = *reinterpret_cast<const floatx4 *>(sv_p[1]),
                          s2 = *reinterpret_cast<const floatx4 *>(sv_p[2]);
            sq0[0] = s0[0]; sq0[1] = s1[0]; sq0[2] = s2[0];
#pragma unroll
            for (int q = 0; q < HQ; q++) zp[q] = pv[0][q] + pv[1][q] + pv[2][q];
            if (GEN && (norm == 0 || norm == 2)) {    // no normaliser: the guard wants the quietest column of the window; Normalize: the window's minimum
                ssp = fminf(fminf(g < T ? s0[0] : INFINITY, g + 4 < T ? s1[0] : INFINITY), g + 8 < T ? s2[0] : INFINITY);
                st1p = norm == 0 ? s0[2] + s1[2] + s2[2]     // (... and the window's whole sum of squares: rows of taps past timeRange read zeros)
                                 : fmaxf(fmaxf(g < T ? s0[2] : -INFINITY, g + 4 < T ? s1[2] : -INFINITY), g + 8 < T ? s2[2] : -INFINITY);
            } else {                                  // sums (rows of taps past timeRange read zeros)
                ssp = s0[0] + s1[0] + s2[0];
                st1p = s0[2] + s1[2] + s2[2];
            }
            fwp = fmaxf(fmaxf(s0[1], s1[1]), s2[1]);
        }
        float zt[HQ], ssw, fww, st1w = 0.0f;
        {
#pragma unroll
            for (int q = 0; q < HQ; q++) {
                auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(zp[q][0]), __float_as_uint(zp[q][1]), false, false);
                const float s01 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                r = __builtin_amdgcn_permlane16_swap(__float_as_uint(zp[q][2]), __float_as_uint(zp[q][3]), false, false);
                const float s23 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
                zt[q] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
            auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(ssp), __float_as_uint(ssp), false, false);
            if (GEN && (norm == 0 || norm == 2)) {
                const float m = fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
                r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                ssw = fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
            } else {
                ssw = xor32_sum(__uint_as_float(r[0]) + __uint_as_float(r[1]));
            }
            if (GEN && norm == 2) {                   // the window's maximum
                r = __builtin_amdgcn_permlane16_swap(__float_as_uint(st1p), __float_as_uint(st1p), false, false);
                const float m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
                r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                st1w = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
            } else if (GEN && norm == 0) {            // the window's sum of squares (relative to the reference exponent)
                st1w = xor32_sum(xor16_sum(st1p));
            } else if (GEN && norm == 3) {
                // mean and M2 of the window from its frames' (equal counts F): M2 = sum M2_t + F sum (mean_t - mean)^2
                const float mean = ssw / (float)T;
                float dv = 0.0f;
#pragma unroll
                for (int tt = 0; tt < 3; tt++) {
                    const float dl = sq0[tt] - mean;
                    dv = g + 4 * tt < T ? fmaf(dl, dl, dv) : dv;
                }
                st1w = xor32_sum(xor16_sum(st1p)) + (float)d.F * xor32_sum(xor16_sum(dv));
                ssw = mean;
            }
            r = __builtin_amdgcn_permlane16_swap(__float_as_uint(fwp), __float_as_uint(fwp), false, false);
            const float m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
            r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            fww = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
        }
        // the rest of the network (NeuralNet.swift:47-59 L2Normalize on the folded first layer, :189-194 TanSig, :366-377 second
        // layer, :137-142 / :175-180 reverse output map; SyllableDetector.swift:27-31 threshold)
        float yv, gstat = ssw;                       // (gstat: what the guard holds against the grid's floor)
        bool hit;
        {
            const int ush = (GEN && scaling != 0) ? 0 : d.col_shift - se_ref;
            const float alpha0 = d.w_unscale * pow2f(ush < -120 ? -120 : (ush > 120 ? 120 : ush));
            float alpha = norm == 1 ? d.w_unscale * __builtin_amdgcn_rsqf(ssw) : alpha0, beta = 0.0f;
            if (GEN && norm == 2) {                   // Normalize: 2 (v - mn) / (mx - mn) - 1, all -1 when mx == mn
                const float range = st1w - ssw;
                gstat = range;
                if (range == 0.0f) { alpha = 0.0f; beta = -1.0f; }
                else { alpha = d.w_unscale * 2.0f / range; beta = (0.0f - ssw - st1w) / range; }
            } else if (GEN && norm == 3) {            // NormalizeStd: (v - mean) / sigma, population sigma
                const float sdv = sqrtf(st1w / (float)d.I);
                gstat = sdv;
                alpha = d.w_unscale / sdv;
                beta = -ssw / sdv;
            }
            float b0_l[HQ], w1_l[HQ], w1o_l[4][HQ], rv_l[HQ];
            if (kTbl) {
                const float *tp = gtab + 8;
                asm volatile("" : "+v"(tp));
#pragma unroll
                for (int q = 0; q < HQ; q++) {
                    b0_l[q] = tp[q];
                    w1_l[q] = tp[HQ + q];
                    rv_l[q] = norm >= 2 ? tp[6 * HQ + q] : 0.0f;
#pragma unroll
                    for (int o = 0; o < 4; o++) w1o_l[o][q] = multi ? tp[2 * HQ + o * HQ + q] : 0.0f;
                }
            } else {
#pragma unroll
                for (int q = 0; q < HQ; q++) {
                    b0_l[q] = b0g[q];
                    w1_l[q] = w1g[q];
                    rv_l[q] = rvg[q];
#pragma unroll
                    for (int o = 0; o < 4; o++) w1o_l[o][q] = w1o[o][q];
                }
            }
            float act[HQ];
#pragma unroll
            for (int q = 0; q < HQ; q++)
                act[q] = transfer_fn(tf0, (GEN && norm >= 2) ? fmaf(alpha, zt[q], fmaf(beta, rv_l[q], b0_l[q])) : fmaf(alpha, zt[q], b0_l[q]));
            float ysum;
            if (multi) {
                float yp[4];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    yp[o] = w1o_l[o][0] * act[0];
#pragma unroll
                    for (int q = 1; q < HQ; q++) yp[o] = fmaf(w1o_l[o][q], act[q], yp[o]);
                }
                auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(yp[0]), __float_as_uint(yp[1]), false, false);
                const float a01 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                r = __builtin_amdgcn_permlane16_swap(__float_as_uint(yp[2]), __float_as_uint(yp[3]), false, false);
                const float a23 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a01), __float_as_uint(a23), false, false);
                ysum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            } else {
                float yq = w1_l[0] * act[0];
#pragma unroll
                for (int q = 1; q < HQ; q++) yq = fmaf(w1_l[q], act[q], yq);
                ysum = xor32_sum(xor16_sum(yq));
            }
            float ob1 = c_b1, ooa = lean_oa, oog = lean_og, oob = lean_ob;
            double othr = thr_g;
            if (kTbl) {
                const float *tp = gtab + 22;
                asm volatile("" : "+v"(tp));
                const floatx4 ov = *reinterpret_cast<const floatx4 *>(tp - 2);      // [.., .., b1, oa]
                ob1 = ov[2]; ooa = ov[3];
                oog = tp[2]; oob = tp[3];
                othr = *reinterpret_cast<const double *>(tp + 4);
            }
            float y = transfer_fn(tf1, ysum + ob1);
            y = (y - ooa) / oog + oob;
            yv = y;
            hit = counts && (double)y >= othr;
            if (multi) {
                unsigned hb = hit ? 1u : 0u;
                auto r = __builtin_amdgcn_permlane16_swap(hb, hb, false, false);
                hb = r[0] | r[1];
                r = __builtin_amdgcn_permlane32_swap(hb, hb, false, false);
                hit = (r[0] | r[1]) != 0u;
            }
        }
        const int er = kTile * t - (T - 1) + n;       // evaluation index inside the segment
        const bool vld = er >= 0 && er < seg_len;
        {
            const bool st = vld && g == 0, sto = vld && g < n_out;
            const unsigned off = e_b32 + (unsigned)er;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yv), out_rs, sto ? (off * (unsigned)n_out + (unsigned)g) * 4u : 0xFFFFFFFFu, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(hit ? 1 : 0), flg_rs, st ? off : 0xFFFFFFFFu, 0, 0);
        }
        // ---- the precision guard (kernels.hpp, FixItem): the window statistic against the loudest grid floor among its frames
        if (guard_on) {
            // (all frames silent: exact zeros, the fused result is the reference's 0/0; a frame the grid cannot hold: +inf, nothing passes)
            bool bad = vld && !(fww == 0.0f) && !(gstat >= guard_k * ((GEN && norm >= 2) ? sqrtf(fww) : fww));
            if (GEN && norm == 0 && fww != INFINITY && fww != 0.0f) {
                // no normaliser: loud enough for the floor not to matter?  (the floor in true units against the network's sensitivity)
                // fww = 4^(se_ref - se_min): se_min = se_ref - log2(fww) / 2
                const int lg = (int)((__float_as_uint(fww) >> 23) & 0xffu) - 127;
                if (se_ref - lg / 2 >= d.guard_se_abs_r) bad = false;
            }
            if (GEN && norm == 0 && scaling == 0 && vld) {
                // ... and loud enough for the arithmetic's own relative error to matter?  (fused_plan.cpp, guard_loud: the window's
                // norm in true units through the network's root-sum-square gain)
                const int ush2 = 2 * (d.col_shift - se_ref);
                const float u2 = st1w * pow2f(ush2 < -120 ? -120 : (ush2 > 120 ? 120 : ush2));
                if (u2 * d.guard_loud > 1.0f) bad = true;
            }
            if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
                const int er0 = kTile * t - (T - 1);
                const int lo = er0 < 0 ? 0 : er0, hi = er0 + kTile < seg_len ? er0 + kTile : seg_len;
                if (lane == 0 && hi > lo) {
                    const unsigned sl = atomicAdd(d.fix.counters, 1u);
                    if (sl < d.fix.capacity) d.fix.items[sl] = FixItem{c, e_b32 + (unsigned)lo, hi - lo, 0};
                    else d.fix.counters[3] = 1u;
                }
            }
        }
        // ---- the carried rows' writes (their reads went out with the evaluation's)
        {
            floatx4 *dst = reinterpret_cast<floatx4 *>(rows);
            if (HQ == 1) {
                if (lane < nq) dst[lane] = cy0;
                if (lane + 64 < nq) dst[lane + 64] = cy1;
                if (nq > 128 && lane + 128 < nq) dst[lane + 128] = cy2;
            } else {
                const floatx4 *src = reinterpret_cast<const floatx4 *>(rows + kTile * PS);
                for (int i = lane; i < nq; i += 64) dst[i] = src[i];
            }
        }
        fo += (unsigned)(kTile * hop);
        fo = fo >= (unsigned)R ? fo - (unsigned)R : fo;
        if (CS8) {                                    // eight chunks a tile: the slots move on, the pads (q & 7) stay
            sb1 += 8u * kChunkB; sb1 = sb1 >= (unsigned)(RC * kChunkB) ? sb1 - (unsigned)(RC * kChunkB) : sb1;
            sb2 += 8u * kChunkB; sb2 = sb2 >= (unsigned)(RC * kChunkB) ? sb2 - (unsigned)(RC * kChunkB) : sb2;
        }
    }
#ifdef SYLDET_S_STAMPS
    if (d.stamps && lane == 0) {
        atomicAdd(d.stamps + 0, __builtin_amdgcn_s_memtime() - st_begin);
        atomicAdd(d.stamps + 1, st_vm);
        atomicAdd(d.stamps + 2, st_lg);
        atomicAdd(d.stamps + 3, (unsigned long long)tiles);
        atomicAdd(d.stamps + 4, 1ull);
    }
#endif
#undef SD_STAMP
}

template <int K2, bool GEN, int HQ, int NW, int PADP = 0, bool F2 = false, int NT = 1, bool SPECT = false>
hipError_t launch_one(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t s_eff, int64_t E,
                      float *outputs, uint8_t *flags, hipStream_t stream)
{
    auto kern = fused_s_kernel<K2, GEN, HQ, NW, PADP, F2, NT, SPECT>;
    constexpr int kWaves = NW;
#ifdef SYLDET_S_ONEWAVE
    const int lds = kWaves == 4 ? 100 * 1024 : d.s_lds_wave * kWaves;       // (one workgroup a CU)
#else
    const int lds = d.s_lds_wave * kWaves;
#endif
    hipError_t st = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (st != hipSuccess) return st;
    const int64_t segs = (E + d.s_seg_evals - 1) / d.s_seg_evals;         // wave segments per channel
    dim3 grid((unsigned)((segs + kWaves - 1) / kWaves), (unsigned)C);
    hipLaunchKernelGGL(kern, grid, dim3(64 * kWaves), (size_t)lds, stream, d, samples, stride, s_eff, E, outputs, flags);
    return hipGetLastError();
}

}  // namespace

// The class fused_r_kernel takes (two layers, at most 4 hidden units, at most 4 outputs, linear |X| columns, no normaliser
// or l2normalize in front of the affine maps, at most one output map), for windows of 64, 128, 192 or 256 samples that are
// symmetric (all of the reference's are), any timeRange up to 12, hops that are multiples of 4 and leave room for the ring.
bool fused_s_has_stamps()
{
#ifdef SYLDET_S_STAMPS
    return true;
#else
    return false;
#endif
}

bool fused_s_applicable(const FusedDesc &d)
{
    // (log / dB columns too: every frame is transformed at its own scale, so a bin's error is relative to its frame, as an fp32
    // FFT's is -- what the logarithm makes of that is the same for both)
    const bool cls = d.norm >= 0 && d.norm <= 3 && d.n_layers == 2 && d.n_out >= 1 && d.n_out <= 4 && d.H <= 16 && d.n_out_fns <= 1;
    if (d.F > 32 && !(d.s2_ok && d.s2_nt == 2 && !d.no_fold2)) return false;      // (only the twice-folded form holds more than 32 bins)
    return d.s_ok && d.T <= 12 && cls;
}

hipError_t launch_fused_s(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t S, int64_t J,
                          int64_t E, float *outputs, uint8_t *flags, hipStream_t stream)
{
    (void)S;
    if (E <= 0 || C <= 0) return hipSuccess;
    if (!fused_s_applicable(d)) return hipErrorInvalidValue;
    const int64_t s_eff = (J - 1) * (int64_t)d.hop + d.gap + d.W;
    if (s_eff * 4 >= 0x7fffffffll) return hipErrorInvalidValue;           // (the launcher's caller keeps such rows on the other kernels)
    const bool exact = d.norm == 1 && d.scaling == 0 && d.tf0 == 0 /* TanSig */ && d.tf1 == 2 /* PureLin */ && d.n_out == 1;
#define SD_S_GO(K2_)                                                                                                  \
    if (d.W == 64 * K2_) {                                                                                            \
        if (d.H > 12) return launch_one<K2_, true, 4, 4>(d, samples, stride, C, s_eff, E, outputs, flags, stream);    \
        if (d.H > 8) return launch_one<K2_, true, 3, 4>(d, samples, stride, C, s_eff, E, outputs, flags, stream);     \
        if (d.H > 4 && d.s_waves == 8) return launch_one<K2_, true, 2, 8>(d, samples, stride, C, s_eff, E, outputs, flags, stream); \
        if (d.H > 4) return launch_one<K2_, true, 2, 4>(d, samples, stride, C, s_eff, E, outputs, flags, stream);     \
        if (exact) return launch_one<K2_, false, 1, 8>(d, samples, stride, C, s_eff, E, outputs, flags, stream);      \
        return launch_one<K2_, true, 1, 8>(d, samples, stride, C, s_eff, E, outputs, flags, stream);                  \
    }
    // hops that are multiples of 64: the padded ring (256-sample windows, up to 4 hidden units: fused_plan.cpp)
#define SD_S_PAD(K2_, P_)                                                                                              \
    if (d.W == 64 * K2_ && d.s_padp == P_) {                                                                           \
        if (d.H > 4) return hipErrorInvalidValue;                                                                      \
        if (K2_ == 4 && d.s2_ok && !d.no_fold2) {                                                                      \
            if (exact) return launch_one<4, false, 1, 8, P_, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream); \
            return launch_one<4, true, 1, 8, P_, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);       \
        }                                                                                                              \
        if (exact) return launch_one<K2_, false, 1, 8, P_>(d, samples, stride, C, s_eff, E, outputs, flags, stream);   \
        return launch_one<K2_, true, 1, 8, P_>(d, samples, stride, C, s_eff, E, outputs, flags, stream);               \
    }
    // hop 128 under the twice-folded form: whole chunks staggered over the banks (CS8; SYLDET_FUSED_PAD128=1 keeps the padded pieces)
    if (d.s_cs8 && !d.no_cs8 && d.s2_ok && !d.no_fold2 && d.W == 256 && d.H <= 4 && d.s2_nt == 1) {
        if (exact) return launch_one<4, false, 1, 8, 1, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
        return launch_one<4, true, 1, 8, 1, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
    }
    if (d.s_padp) {
        SD_S_PAD(4, 64) SD_S_PAD(4, 128)
        return hipErrorInvalidValue;
    }
#undef SD_S_PAD
    // W == N == 256, up to four hidden units: the twice-folded instantiation (SYLDET_FUSED_NOFOLD2=1 keeps the once-folded one: A/B runs)
    if (d.s2_ok && d.s2_nt == 2) {                   // bands of 33 .. 64 bins: the twice-folded form with two row tiles per parity, 4 waves
        if (d.no_fold2 || d.s_padp) return hipErrorInvalidValue;
        if (exact) return launch_one<4, false, 1, 4, 0, true, 2>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
        return launch_one<4, true, 1, 4, 0, true, 2>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
    }
    if (d.s2_ok && !d.no_fold2 && d.W == 256) {       // (5 .. 16 hidden units too: the halved basis and the single pass do not depend on the layer's width)
        if (d.H > 12) return launch_one<4, true, 4, 4, 0, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
        if (d.H > 8) return launch_one<4, true, 3, 4, 0, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
        // (5 .. 8 hidden units on eight waves stay once-folded: with two accumulator sets AND the kept samples the twice-folded
        // form spills at 256 registers and measures 10 % slower there -- except under hops that are multiples of 64, where
        // reading the ring once (its rows then collide on the banks: there is no padded ring for wider layers) is worth 15 %;
        // MEASUREMENTS.md R4.8)
        if (d.H > 4 && d.s_waves == 8 && d.hop % 64 == 0) return launch_one<4, true, 2, 8, 0, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
        if (d.H > 4 && d.s_waves == 8) return launch_one<4, true, 2, 8>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
        if (d.H > 4) return launch_one<4, true, 2, 4, 0, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
#ifdef SYLDET_S_ONEWAVE                // (diagnostic build: ONE wave a SIMD -- how long does a tile take a wave with nobody to share with?)
        if (exact) return launch_one<4, false, 1, 4, 0, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
#endif
        if (exact) return launch_one<4, false, 1, 8, 0, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
        return launch_one<4, true, 1, 8, 0, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
    }
    SD_S_GO(4) SD_S_GO(2) SD_S_GO(1) SD_S_GO(3)
#undef SD_S_GO
    return hipErrorInvalidValue;
}

// The DFT front half alone on the twice-folded form: |X| or |X|^2 columns [C][J][F] (d.spect_out), for 256-point frames under a
// 256-sample window; the plan is the stand-in network's (timeRange 1, one unit: syldet_api.cpp, build_dft_plan).
bool fused_s_spectrogram_applicable(const FusedDesc &d)
{
    return d.s_ok && d.s2_ok && d.s2_nt == 1 && !d.no_fold2 && d.W == 256 && d.T == 1 && d.F <= 32 && (d.s_padp == 0 || d.s_padp == 64 || d.s_padp == 128);
}

hipError_t launch_fused_s_spectrogram(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t J, hipStream_t stream)
{
    if (J <= 0 || C <= 0) return hipSuccess;
    if (!fused_s_spectrogram_applicable(d) || d.spect_out == nullptr) return hipErrorInvalidValue;
    const int64_t s_eff = (J - 1) * (int64_t)d.hop + d.gap + d.W;
    if (s_eff * 4 >= 0x7fffffffll || (uint64_t)J * (uint64_t)d.F * 4u >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    if (d.s_cs8 && !d.no_cs8) return launch_one<4, false, 1, 8, 1, true, 1, true>(d, samples, stride, C, s_eff, J, nullptr, nullptr, stream);
    if (d.s_padp == 64) return launch_one<4, false, 1, 8, 64, true, 1, true>(d, samples, stride, C, s_eff, J, nullptr, nullptr, stream);
    if (d.s_padp == 128) return launch_one<4, false, 1, 8, 128, true, 1, true>(d, samples, stride, C, s_eff, J, nullptr, nullptr, stream);
    return launch_one<4, false, 1, 8, 0, true, 1, true>(d, samples, stride, C, s_eff, J, nullptr, nullptr, stream);
}

}  // namespace sd
