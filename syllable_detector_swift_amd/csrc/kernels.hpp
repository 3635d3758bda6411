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
// the same for 128-, 256- and 512-point frames, butterflies across the lanes (kernels_stft_lanes.hip)
bool stft_lanes_applicable(const StftDesc &d, const float *samples, int64_t stride);
hipError_t launch_stft_lanes(const StftDesc &d, const float *samples, int64_t stride, int C, int64_t J, float *columns, hipStream_t stream);
// outputs [C][E][n_out], flags [C][E] <- columns [C][J][F]; either output may be null
hipError_t launch_mlp_generic(const NetDesc &n, int F, const float *columns, int C, int64_t J, int64_t E,
                              float *outputs, uint8_t *flags, hipStream_t stream);
// indices [C][capacity], counts [C] <- flags [C][E]
hipError_t launch_detections(const uint8_t *flags, int C, int64_t E, int64_t first_index, int64_t hop,
                             int64_t debounce_frames, int64_t *indices, int64_t capacity, int64_t *counts,
                             hipStream_t stream);

// ---- wide-network engine (kernels_wide.hip): first layer as a bf16 MFMA GEMM over many evaluations ----
constexpr int kWideBlock = 1024;           // 16 waves, 32 evaluations each
constexpr int kWideTile = 512;             // evaluations per workgroup
constexpr int kWideK = 320;                // inputs per evaluation, zero padded (20 k-steps of 16)
constexpr int kWideChunkBytes = 20 * 64 * 16 + (32 + 4 * 32) * 4;   // 32 hidden units: A fragments, then b0[32], w1[4][32]

struct WideDesc {
    int H, n_chunks;            // first-layer outputs, chunks of 32 of them (zero padded)
    int n_out, tf0, tf1, rule, n_out_fns;
    int sig;                    // TanSig / LogSig hidden layer folded into the tables (see wide_gemm_kernel)
    int m32;                    // the staggered two-workgroup GEMM on v_mfma_f32_32x32x16_bf16, chunks packed for it: SYLDET_WIDE_M32 (kernels_wide.hip, wide_gemm32s_kernel)
    int poly;                   // (sig, shape16) the folded tables are for tanh_poly(acc) instead of 1 / (2^acc + 1): SYLDET_WIDE_TANH_POLY
    int shape16;                // the chunks are packed for v_mfma_f32_16x16x32_bf16 ([k-step of 32][unit tile of 16]): wide_gemm16_kernel
    // front = 1: the GEMM kernel reads the |X| columns itself (an evaluation's inputs are I consecutive floats of [C][J][F]) --
    // the input chain is [l2normalize,] affine maps on linear columns, the affine part folded into the first layer by the host,
    // so the B operands are bf16(v / |v|) (l2 = 1) or bf16(v): no [evaluations][320] image in HBM, no preparation kernel
    int front, l2, I, F;
    int wg8;                    // two workgroups of 8 waves a CU instead of one of 16 (kernels_wide.hip, NWV)
    int stagger;                // (wg8) waves 4-7 run one epilogue behind waves 0-3 (kernels_wide.hip, STG)
    int dma_builtin;            // (wg8, unstaggered) the weight DMA through __builtin_amdgcn_global_load_lds instead of the assembly statement
    int tiles4;                 // (wg8, staggered) one workgroup of 8 waves a CU with four evaluation tiles a wave (kernels_wide.hip, TPW)
    const uint4 *wpack;         // [n_chunks][kWideChunkBytes / 16]
    const float *b1;            // [n_out]
    const float *out_params;    // per output fn: y, gain[n_out], xoff[n_out]
    const double *thresholds;   // [n_out]
};
// columns [C][J][F] -> xn [C*E][kWideK] bf16 (scaling + input functions applied)
int wide_front_stage_floats(int F, int I);
bool wide_front_fits(int F, int I);        // WideDesc::front needs the columns under 512 evaluations in LDS
bool wide_prep_is_chain(const NetDesc &n);   // which of the two preparation kernels launch_wide_prep picks (for timing labels)
hipError_t launch_wide_prep(const NetDesc &n, int F, const float *columns, int C, int64_t J, int64_t E, void *xn, hipStream_t stream);
// xn [NE][kWideK] -> outputs [NE][n_out], flags [NE]
// (front: columns [C][J][F], E evaluations per channel, NE = C E; xn unused)
hipError_t launch_wide_gemm(const WideDesc &d, const void *xn, const float *columns, int64_t J, int64_t E, int64_t NE, float *outputs,
                            uint8_t *flags, hipStream_t stream);

// ---- fused engine (kernels_fused.hip) ----------------------------------------------------
// One kernel: samples -> outputs + flags.  The band-limited windowed DFT of 32 frames at a
// time is a GEMM on the matrix cores (f16 hi/lo split operands, fp32 accumulate), the first
// network layer is folded into a second MFMA on the accumulator tile, the sliding window is a
// diagonal sum over an LDS ring.  Built by make_fused_plan() when the configuration fits.
constexpr int kFusedBlock = 512;        // 8 waves
constexpr int kFusedChunkFrames = 64;   // frames per team chunk (4 waves x 16); a workgroup = 2 teams
constexpr int kFusedTileFrames = 128;   // frames per workgroup period (two chunks)
constexpr int kFusedMaxLoads = 10;      // float4 loads per thread per pass
constexpr int kFusedRBlock = 256;       // register-resident-basis kernel (kernels_fused_r.hip): 4 waves, one per SIMD
constexpr int kFusedRTileFrames = 64;   //   16 frames per wave and pass
constexpr int kFusedRMaxLoads = 10;     //   float4 loads per thread per pass
constexpr int kFusedRPRows = 11 + 3 * 64 + 2;   //   rows of its tap-product ring in LDS (52 floats each)
constexpr int kFusedSBlock = 512;       // symmetric-fold kernel (kernels_fused_s.hip): 8 waves, two per SIMD, each a stream of its own
constexpr int kFusedSTileFrames = 16;   //   frames per wave and tile
constexpr int kFusedColStride = 40;     // halves per column row in LDS: 32 bins + 8 of padding (80 B rows: 16-byte
                                        // aligned, and 16 consecutive rows cover all 64 banks once for ds_read_b128)

// ---- precision guard + exact recomputation (kernels_fixup.hip) ----------------------------
// The fused kernels are block floating point: a pass (64 or 128 frames) is scaled by its loudest sample and split into
// f16 hi + lo.  A window that sits far below that level (a quiet stretch next to a click), a pass with an infinite sample
// or a level step of hundreds of dB cannot be held on that grid.  The kernels know: an evaluation whose window norm,
// measured on the grid it was computed on, is too close to the grid's floor for the network's sensitivity (a host-side
// Lipschitz bound) is reported in a work list, and fixup_kernel recomputes exactly those evaluations from the samples
// (fp64 DFT by definition, the network unfolded in the reference's operation order) behind the fused kernel on the same
// stream.  NaN may then appear only where the reference's windows contain the offending sample.
struct FixItem {
    int c;                      // channel
    unsigned first;             // first evaluation (kind 0) / frame (kind 1: spectrogram columns)
    int count;
    int kind;
};
struct FixList {
    unsigned *counters;         // [0] items appended, [1] workgroups done, [2] items of the last launch, [3] overflow (sticky), [4] the launch's first 10 ns tick (of 8 words)
    FixItem *items;
    unsigned capacity;
    unsigned *host_count = nullptr;   // two page-locked words for the launch's item count and its duration in 10 ns ticks (profiling: syldet_timings), or null
};

struct FusedDesc {
    int W, KS;                  // window length, k-steps of 32 samples (KS*32 >= W)
    int hop, gap, F, T;         // frame advance, leading gap, bins, timeRange
    int H;                      // first-layer outputs
    int norm;                   // 0 none, 1 l2normalize, 2 normalize, 3 normalizestd (first input fn)
    int scaling;
    int n_layers, n_out, tf0, tf1, rule, n_out_fns;
    int I;                      // F*T
    int nsmp, nload;            // samples staged per pass, float4 loads per thread
    int skew;                   // floats of padding after every `hop` staged samples (bank spreading)
    unsigned hop_magic;         // ceil(2^32 / hop): i / hop == umulhi(i, hop_magic) for i < 2^16
    int runs, seg_evals;        // passes per workgroup, evaluations per workgroup segment
    int ps;                     // column slots: 2 (T - 1) for the transition strip + 128 for the pass
    int smp_stride;             // halves between the hi and the lo array of the staged samples
    int col_shift;              // |X| columns are stored as |X| * 2^(cse - col_shift) (2x for |X|^2)
    float w_unscale;            // 1 / (power-of-two scale of the folded first-layer weights)
    int lds_dfrag, lds_smp, lds_colh, lds_coll, lds_stat, lds_red, lds_cst, lds_total;   // byte offsets
    // layout of the register-resident-basis kernel (64-frame passes, staged samples double-buffered, no basis in LDS);
    // r_ok = 0 when the shape does not fit it (long hops)
    int classic_ok;             // the 8-wave kernel's LDS layout fits (else only the register-resident-basis kernel can run the plan)
    int r_ok, r_nsmp, r_nload, r_ps, r_smp_stride, r_runs, r_seg_evals;
    int r_lds_smp, r_lds_p, r_lds_red, r_lds_cst, r_lds_total;   // second sample buffer: r_lds_smp + 4 r_smp_stride
    // layout of the symmetric-fold kernel (kernels_fused_s.hip): per wave a ring of s_ring_chunks x 256 samples + a mirror chunk,
    // rows of s_pstride floats of tap products (4 s_tp products, the frame's sum of squares, its floor weight); s_ok = 0 when the
    // shape does not fit it
    int s_ok, s_waves, s_perm, s_ring_chunks, s_pstride, s_tp, s_lds_wave, s_seg_evals;
    int s_padp;                 // 0, or the padding period (floats) of the fold kernel's sample ring: hops that are multiples of 64 (fused_plan.cpp)
    // ... and its second-fold instantiation (W == N == 256, one quad of units, plain ring): s2_ok; s2_pe / s2_po = band index of
    // the first even / odd bin (0 and 1 in some order): lane group g of a result holds band bins 8 g + s2_pe + 2 i (even tile
    // rows 4 g + i) and 8 g + s2_po + 2 i (odd tile)
    int s2_ok, s2_pe, s2_po;
    int s2_nt;                  // row tiles per parity of the twice-folded form: 1 (up to 32 bins) or 2 (33 .. 64 bins: 4 waves a workgroup)
    int s_cs8, s_cs8_rc;        // hop 128 under a 256-sample window, one quad of units: the ring as whole chunks staggered over the banks (kernels_fused_s.hip, CS8) and its slots
    int no_cs8;                 // the handle was created under SYLDET_FUSED_PAD128=1: the padded pieces where both take the shape (A/B runs)
    int no_fold2;               // the handle was created under SYLDET_FUSED_NOFOLD2=1: the once-folded form where both take the shape (A/B runs)
    const uint4 *sfrag2;        // [2 k-steps][Re even, Re odd, Im even, Im odd][hi,lo][64 lanes] A-operand fragments of the twice-folded basis
    const float *swin2;         // [4 lane groups][2 k-steps][8][2] window coefficients w[128 + m], w[m] for m = 32 ks + 8 g + i
    const float *s2c;           // [64 lanes][8]: cos(pi k / 2) 2^13 for the lane's four even bins, w[192], padding
    const uint4 *afrag_t2;      // afrag_t in the bin order of the twice-folded result
    const uint4 *afrag_w2;      // afrag_w in that order (5 .. 16 hidden units)
    const uint4 *sfrag;         // [W/64 k-steps][s, d][bins 0-15, 16-31][hi,lo][64 lanes] A-operand fragments of the folded basis
    const uint4 *afrag_w;       // [3][HQ quads of hidden units][hi,lo][64 lanes] the first layer with all taps as rows for 5 .. 16 units
    const float *slone;         // [64 lanes][8] the frame's first sample's real coefficients for the lane's bins
    const uint4 *dfrag;         // [KS][re 0-15, re 16-31, im 0-15, im 16-31][hi,lo][64 lanes] A-operand fragments of the DFT basis
    const uint4 *afrag;         // [T][hi,lo][64 lanes] A-operand fragments of the folded first layer (16x16x32)
    const uint4 *afrag_t;       // [3 row tiles][hi,lo][64 lanes] the same layer with all taps as rows (row 4 t + h), K in the
                                // bin order of a magnitude result: k = 8 g + j -> bin 4 g + j (j < 4), 16 + 4 g + j - 4 (j >= 4);
                                // null unless H <= 4
    const int *koff;            // [KS][4] staged-sample offset of k-step ks for lane group g4 (skew applied)
    const float *bias0;         // [H]  b0 + W0 . (constant part of the input maps)
    const float *rvec;          // [H]  (W0 o a) . 1
    const float *w1, *b1;       // layer 1, row-major [n_out][H] (2-layer nets)
    const float *out_params;    // per output fn: y, gain[n_out], xoff[n_out]
    const double *thresholds;   // [n_out]
    float *spect_out;           // spectrogram instantiation only: [C][J][F] columns
    int spect_power;            //   0: |X|, 1: |X|^2
    // precision guard (see FixItem): thresholds on the window statistic in grid units, from fused_plan.cpp
    float guard_r;              // register-resident-basis kernel: window sum of squares, times 4^(se_ref - se_min)
    float guard_c;              // 8-wave kernel, l2normalize / no normaliser: window sum of squares
    float guard_c_range;        //   normalize: window range;  normalizestd: window sigma
    float guard_range_r;        // symmetric-fold kernel, normalize: window range; normalizestd: window sigma (relative to the loudest frame's floor)
    float guard_rel_r, guard_rel_c;   // no normaliser: the per-column relative criterion (smallest column sum of squares of the
                                      // window) that joins the absolute one
    int guard_se_abs_r, guard_se_abs_c;   // no normaliser: passes scaled below this exponent are loud enough for the floor to matter
    float guard_spect;          // spectrogram instantiation: a frame's column sum of squares (grid units) ...
    int guard_se_abs_s;         //   ... in passes scaled below this exponent
    float guard_loud;           // no normaliser: (window sum of squares in true units) * guard_loud > 1 means the three-product arithmetic's
                                //   2^-21.4 of the column level, through the network, is expected beyond a quarter of the 1e-5 bar (symmetric-fold kernel)
    FixList fix;                // work list of evaluations to recompute (null counters: guard off)
    int force_classic;          // the handle was created under SYLDET_FUSED_CLASSIC=1: the 8-wave kernel where both take the shape
    int no_fold;                // the handle was created under SYLDET_FUSED_NOFOLD=1: not the symmetric-fold kernel
    int ko;                     // diagnostic build only: knock-out mask (SYLDET_FUSED_KO)
    unsigned long long *stamps; // diagnostic build only: [workgroups][16] phase cycle sums, else null
};

// ---- first layer on the matrix cores for spectrograms already in HBM (kernels_mlpx.hip) --------------------
// The generic engine's network stage for the detector class the training script writes (l2normalize first, affine
// maps, TanSig hidden layer of at most 4 units, one linear output, at most one output map, linear |X| columns) when the
// band is too wide or the window too long for the fused engine (BASELINE configs[2]: 116 bins of 1024-point frames):
// the fused engine's shifted GEMM over f16 hi/lo columns, fed from [C][J][F] columns instead of from its own DFT.
constexpr int kMlpxBlock = 512;          // 8 waves, 16 frames each
constexpr int kMlpxTile = 128;           // frames per tile (its windows: 128 - timeRange + 1 evaluations)
struct MlpxDesc {
    int F, T, KB, H;            // bins, timeRange, 32-bin blocks per tap (F <= 32 KB), hidden units
    int rule;
    int scaling;                // SYLDET_SCALING_*: linear, ln x or 20 log10 x of the columns in front of the chain (SyllableDetector.swift:184-212)
    int col_stride;             // halves per column row in LDS: 32 KB + 8
    int p_stride;               // floats per frame row of tap products in LDS
    float w_unscale, b1, oa, og, ob;   // 1 / scale of the folded weights; second layer bias; output map (y - oa) / og + ob
    int lds_afrag, lds_colh, lds_coll, lds_p, lds_pq, lds_ss, lds_red, lds_total;   // byte offsets
    const uint4 *afrag;         // [3 row tiles][KB][hi,lo][64 lanes] A-operand fragments: row 4 t + h = tap t, hidden unit h
    const float *bias0, *w1;    // [4] folded first-layer biases, second-layer weights (zero padded)
    const double *thresholds;   // [1]
};
// ---- frames of whole hops (W = N = R hop, R = 4, 2 or 1): every block of `hop` samples transformed once on the matrix cores, frames as sliding
// sums of R blocks, the window as three taps along the bins, then the matrix-core network stage -- one launch (kernels_bdft.hip)
constexpr int kBdftBlock = 512;          // 8 waves, two per SIMD; wave w owns bins kb0 + 16 w .. + 15
struct BdftDesc {
    int hop, kb0, f0;           // hop = N / R (128 or 256); first of the 128 bins transformed (a multiple of 4, <= f0 - 1); the band's first bin
    int R;                      // blocks per frame: 4, 2 or 1 (75 %, 50 %, no overlap)
    float a0, a1c, a1s;         // the window as a cosine sum: a_0, (a_1 / 2) cos(pi hop / N), (a_1 / 2) sin(pi hop / N)
    const uint4 *basis;         // [8 waves][cosine rows, sine rows][hop / 64 k-steps][hi,lo][64 lanes] A-operand fragments of the folded block basis
    const float *cre;           // [8 waves][64 lanes][4] the block's first sample's real coefficients for the lane's bins
    const uint4 *afrag;         // [3 row tiles][4][hi,lo][64 lanes] first layer, all taps as rows, K = bin - kb0
};
// outputs [C][E][1], flags [C][E] <- samples [C][stride]; S: samples per channel
hipError_t launch_bdft_net(const MlpxDesc &d, const BdftDesc &bd, const float *samples, int64_t stride, int C, int64_t S, int64_t J, int64_t E,
                           float *outputs, uint8_t *flags, hipStream_t stream);
// ---- 1024-point frames: packed real FFT + the matrix-core network stage in one launch (kernels_fft1k.hip) ----
constexpr int kFft1kBlock = 768;         // 12 waves (three per SIMD): a frame per wave at a time, 10 or 11 frames of a 128-frame tile each
bool fft1k_applicable(const StftDesc &s, const MlpxDesc &d, const float *samples, int64_t stride);
// outputs [C][E][1], flags [C][E] <- samples [C][stride]
hipError_t launch_fft1k_net(const StftDesc &s, const MlpxDesc &d, const float *samples, int64_t stride, int C, int64_t J, int64_t E,
                            float *outputs, uint8_t *flags, hipStream_t stream);
// outputs [C][E][1], flags [C][E] <- columns [C][J][F]
hipError_t launch_mlpx(const MlpxDesc &d, const float *columns, int C, int64_t J, int64_t E, float *outputs, uint8_t *flags,
                       hipStream_t stream);

// ResamplerLinear (Common/Resampler.swift:36-69) for C channels at once; `last` is the per-channel carry on the device
hipError_t launch_resample_linear(const float *in, int64_t n_in, int64_t in_stride, float *out, int64_t n_out,
                                  int64_t out_stride, int C, float step, float offset, float *last, hipStream_t stream);
// whole-recording rate conversion, fp64 positions (offline input: no carry)
hipError_t launch_convert_rate(const float *in, int64_t n_in, int64_t in_stride, float *out, int64_t n_out, int64_t out_stride,
                               int C, double step, hipStream_t stream);
// frame-major [n_frames][total] -> channel-major rows of channels first .. first+C-1
hipError_t launch_deinterleave(const float *in, int64_t n_frames, int total, int first, int C, float *out,
                               int64_t out_stride, hipStream_t stream);

// detection flags <-> bits (bit b of byte t of a row = flag 8 t + b), rows padded to whole bytes
hipError_t launch_pack_flags(const uint8_t *flags, int64_t rows, int64_t row_len, uint8_t *bits, hipStream_t stream);
hipError_t launch_unpack_flags(const uint8_t *bits, int64_t rows, int64_t row_len, uint8_t *flags, hipStream_t stream);
hipError_t launch_unpack_flags_gathered(const uint8_t *bits, int64_t rows, int64_t row_len, int64_t shards, int64_t padded,
                                        uint8_t *flags, hipStream_t stream);
// (the copy exchange: every shard's packed rows behind a pointer of its own, read where they lie)
constexpr int kMaxFlagSources = 16;
struct FlagSources {
    const uint8_t *p[kMaxFlagSources] = {};
};
hipError_t launch_unpack_flags_from(const FlagSources &from, int64_t rows, int64_t row_len, int64_t shards, int64_t padded,
                                    uint8_t *flags, hipStream_t stream);

hipError_t launch_fused(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t S, int64_t J,
                        int64_t E, float *outputs, uint8_t *flags, hipStream_t stream);
int fused_choice(const FusedDesc &d, int64_t J);    // 0 the 8-wave kernel, 1 the register-resident-basis kernel, 2 the symmetric-fold kernel
// the DFT front half alone: samples -> [C][J][F] columns; d: a plan for timeRange 1 with spect_out / spect_power set
hipError_t launch_fused_spectrogram(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t J, hipStream_t stream);
// ... on the symmetric-fold kernel's twice-folded form, where the plan allows it (256-point frames under a 256-sample window)
bool fused_s_spectrogram_applicable(const FusedDesc &d);
hipError_t launch_fused_s_spectrogram(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t J, hipStream_t stream);
// the same contract on the register-resident-basis kernel; only called when d.r_ok and fused_r_applicable(d)
bool fused_r_applicable(const FusedDesc &d);
bool fused_r_has_stamps();      // built with -DSYLDET_R_STAMPS (phase timing, SYLDET_FUSED_STAMPS=1)
bool fused_s_has_stamps();      // built with -DSYLDET_S_STAMPS: where a wave of the fold kernel waits (SYLDET_FUSED_STAMPS=1)
hipError_t launch_fused_r(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t S, int64_t J,
                          int64_t E, float *outputs, uint8_t *flags, hipStream_t stream);
// the same contract on the symmetric-fold kernel; only called when fused_s_applicable(d)
bool fused_s_applicable(const FusedDesc &d);
hipError_t launch_fused_s(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t S, int64_t J,
                          int64_t E, float *outputs, uint8_t *flags, hipStream_t stream);
// taps the register-resident first-layer fragments are instantiated for (0: timeRange too long)
int fused_taps_max(int T);

// ---- exact recomputation of the evaluations the fused kernels reported (kernels_fixup.hip) ----
struct FixDesc {
    int N, W, hop, gap, f0, F, T;
    int power_mode;             // spectrogram items: 0 |X| (extractPower), 1 |X|^2 (extractMagnitude)
    const float *window;        // [W] the fp32 window table (WindowType.createWindow)
    const double2 *ctab;        // [N] (cos, sin)(2 pi m / N)
};
constexpr int kFixMaxCount = 16;            // evaluations (or frames) per work item
// outputs [C][E][n_out], flags [C][E]: the listed evaluations are overwritten; columns [C][J][F]: the listed frames
hipError_t launch_fixup(const FixDesc &fd, const NetDesc &n, const float *samples, int64_t stride, int64_t J, int64_t E,
                        float *outputs, uint8_t *flags, float *columns, const FixList &list, hipStream_t stream);

}  // namespace sd
