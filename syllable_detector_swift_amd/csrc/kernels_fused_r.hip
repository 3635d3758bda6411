// kernels_fused_r.hip -- the fused engine with the DFT basis resident in registers.
//
// Same path, same arithmetic and same tables as kernels_fused.hip (reference, root relative:
//   extractPower          Common/CircularShortTimeFourierTransform.swift:280-337
//   processFourierData    Common/SyllableDetector.swift:134-151
//   processNewValue       Common/SyllableDetector.swift:153-217
//   NeuralNet.apply       Common/NeuralNet.swift:294-326, :366-377
//   lastDetected          Common/SyllableDetector.swift:27-31),
// a different use of the CU.  A gfx950 wave that is alone on its SIMD owns 512 registers: the 64 KB
// of f16 hi/lo basis fragments (A operands of the windowed band-limited DFT) are exactly 256 of them, so
//   * one workgroup = 4 waves, one per SIMD, 16 frames each: 64-frame passes;
//   * the basis never travels through LDS again -- a k-step fetches two B fragments (staged samples, hi + lo) for its 12
//     MFMAs instead of ten fragments;
//   * the first layer takes ALL taps as the rows of one GEMM (row 4 t + h) whose B operand is the wave's own magnitudes, still
//     in the registers the DFT left them in: 9 MFMAs per 16 frames, no |X| columns in LDS, no transition strip; tap products
//     go to an LDS ring and an evaluation is their diagonal sum;
//   * three passes are in flight and everything that crosses waves crosses a pass boundary:
//
//   matrix block of pass q:  DFT(q) from staged buffer q&1
//                            ||  stage pass q+1 into the other buffer, load pass q+2 into the other staging register set
//                            ||  finish pass q-1 (magnitudes, tap products -> ring)  ||  evaluate pass q-2 (ring)
//                            ||  block maximum of pass q+2
//   barrier
//
//   one scheduling region in which every vector, LDS and memory instruction has its place between two MFMAs of the same wave.
//
// gfx950 only.  wave = 64.

#include <type_traits>

#include "fused_common.hpp"

namespace sd {

namespace {

using namespace fused_dev;

constexpr int kBlock = kFusedRBlock;           // 256 threads = 4 waves, one per SIMD
constexpr int kWaves = kBlock / 64;
constexpr int kPass = kFusedRTileFrames;       // 64 frames per pass = 16 per wave
constexpr int kPStride = 52;                   // floats per frame row of tap products: 48 + the frame's sum of squares + padding
                                               // (208-byte rows: 16 consecutive rows cover all 64 banks once for b128 accesses)
constexpr int kPRows = kFusedRPRows;           // T-1 repeated rows (at most 11) + 3 passes x 64 frames + 2 spare
constexpr int kPLead = 11;                     // ring row 0 sits at row kPLead

// KS: k-steps of 32 samples (the basis takes 32 KS registers); TMAX: largest timeRange; NL: staging quads per thread (all NL are
// always loaded and staged: quads past the pass come back as zeros from the descriptor's bounds check and land in LDS
// words no frame reads); SKEW: staged samples carry bank-spreading padding -- instantiated for hops 16, 32, 64 and 128 (SKEW = the hop), where the
// padding (4 halves after every 128 samples) is a matter of constants; STAMP: diagnostic phase timing.
// GEN: the wider network class -- any transfer functions, with or without l2normalize in front -- as run-time facts; without
// it the network class is the reference's example detector's (kernels_fused.hip's LEAN): l2normalize first, linear |X|
// columns, two layers, TanSig hidden units (at most 4), one output, at most one output map.
template <int KS, int TMAX, int NL, int SKEW, bool STAMP, bool GEN>
__global__ void __launch_bounds__(kBlock, 1)
fused_r_kernel(const FusedDesc d, const float *__restrict__ samples, int64_t stride, int64_t s_eff, int64_t E,
               float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *red = reinterpret_cast<float *>(smem + d.r_lds_red);      // [4 waves] block-max partials
    float *cst = reinterpret_cast<float *>(smem + d.r_lds_cst);
    const int T = d.T;                    // timeRange (taps past it are rows of zeros in the first-layer fragments)

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
    const int H = d.H;
    const int n_out = GEN ? d.n_out : 1;              // (GEN: up to four outputs, each finished by its own lane group)
    const int fl = 16 * wave + f;     // this lane's frame / evaluation slot inside the pass
    const int runs = d.r_runs;

    // staged samples: [buffer 0 hi | buffer 0 lo | buffer 1 hi | buffer 1 lo], r_smp_stride halves each
    _Float16 *smp0 = reinterpret_cast<_Float16 *>(smem + d.r_lds_smp);
    const int buf_halves = 2 * d.r_smp_stride;
    // tap products: a ring of three passes' frames, [kPRows][kPStride] fp32 -- per frame 12 taps x 4 units, then its sum
    // of squares (rows 0 .. T-2 repeat the ring's last T-1 frames, so that a window never wraps; two spare rows at the end)
    float *pbuf = reinterpret_cast<float *>(smem + d.r_lds_p);

    // ---- once per workgroup: constants
    if (tid < 16) reinterpret_cast<double *>(cst + kCstThr)[tid] = tid < n_out ? d.thresholds[tid] : 0.0;
    // the DFT basis: A-operand fragments [k-step][re 0-15, re 16-31, im 0-15, im 16-31][hi, lo], one quad per lane each
    uint32x4 a[KS * 8];
#pragma unroll
    for (int i = 0; i < KS * 8; i++) a[i] = reinterpret_cast<const uint32x4 *>(d.dfrag)[i * 64 + lane];
    // first-layer fragments with ALL taps as rows (row 4 t + h: three 16-row tiles), f16 hi + lo; K = this wave's bin
    // order in a magnitude result (see mag_micro)
    half8 aft[3][2];
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int p = 0; p < 2; p++) aft[m][p] = as_half8(reinterpret_cast<const uint32x4 *>(d.afrag_t)[(m * 2 + p) * 64 + lane]);
    for (int i = tid; i < kPRows * kPStride / 4; i += kBlock) reinterpret_cast<floatx4 *>(pbuf)[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    const float c_b1 = GEN ? (g4 < n_out ? d.b1[g4] : 0.0f) : d.b1[0];   // (GEN: this lane group's output)

    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(
        outputs ? outputs + (int64_t)c * E * n_out : nullptr, 0, outputs ? (int)(E * n_out * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t flg_rs = __builtin_amdgcn_make_buffer_rsrc(flags ? flags + (int64_t)c * E : nullptr, 0, flags ? (int)E : 0, 0x00020000);
    float lean_oa = 0.0f, lean_og = 1.0f, lean_ob = 0.0f;
    if (d.n_out_fns == 1) {                       // reverse map (y - y0) / gain + xoff of this lane group's output: [y0 | gains | xoffs]
        const int o = (GEN && g4 < n_out) ? g4 : 0;
        lean_oa = d.out_params[0]; lean_og = d.out_params[1 + o]; lean_ob = d.out_params[1 + n_out + o];
    }

    // this lane's frame in a staged buffer, and where k-step ks of lane group g4 starts inside it (see kernels_fused.hip)
    // SKEW (hop 128: every frame would start on the same LDS bank): sample i sits at i + 4 (i >> 7), so frame fl starts at
    // 132 fl and lane group g4's blocks at 64 g4 + 4 (g4 >> 1) inside it; a k-step's 8 ks < 64 never crosses a padding
    // (in general, with 4 halves of padding behind every hop staged samples -- a hop that is a multiple of 16 puts every
    // frame of a tile on the same banks -- : sample i at i + 4 (i / hop); for a hop that is a power of two everything below
    // is a matter of constants: frame fl starts at (hop + 4) fl, lane group g4's blocks at 8 KS g4 + 4 (8 KS g4 / hop) inside
    // it, k-step ks at 8 ks + 4 (8 ks / hop) behind that -- 8 ks + the group's start never carries into the next padding)
    // SKEW == 1: the other multiples of 16 (48, 80, 96, 112): 4 halves behind every 16 samples -- frames then start 5/4 hop
    // apart, which spreads them over the banks for every odd multiple of 16 and for 96, and everything is again a matter
    // of constants, the same for all those hops: sample n of a frame at n + 4 (n >> 4)
    const int foff = SKEW == 1 ? fl * (d.hop + d.hop / 4) + 10 * KS * g4
                               : (SKEW ? fl * (SKEW + 4) + 8 * KS * g4 + 4 * ((8 * KS * g4) / SKEW) : fl * d.hop + 8 * KS * g4);
    int ko[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) ko[ks] = 8 * ks + (SKEW == 1 ? 4 * (ks >> 1) : (SKEW ? 4 * ((8 * ks) / SKEW) : 0));

    // raw samples of one pass: quads 4*(tid + 256 k), k < NL, through a bounds-checked descriptor
    // two sets: pass q+1 (being staged during the matrix block of pass q) is in set (q+1)&1, pass q+2 arrives in set q&1
    uint32x4 v0[NL], v1[NL];
    auto pass_rsrc = [&](int p) {
        return tile_rsrc(row, (e_b + (int64_t)kPass * p) * d.hop + d.gap, p < runs ? s_eff : 0, d.r_nsmp);
    };
    auto max_partial = [&](const uint32x4 (&v)[NL]) {
        float amax = 0.0f;
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const floatx4 q = as_floatx4(v[k]);
            amax = absmax3(absmax3(amax, q[0], q[1]), q[2], q[3]);
        }
        amax = wave_max_nonneg(amax);
        if (lane == 0) red[wave] = amax;
    };
    // (status of the pass for the precision guard: 0 fine, 1 silent -- all samples zero, its columns are exact zeros --, 2 the
    // grid cannot hold it: an infinite sample, or a level above 2^113; a level below 2^-100 only loses headroom, which the
    // guard's own criterion sees)
    auto scale_of = [&](floatx4 r0, int &status) {
        const float amax = fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[2], r0[3]));
        const int ex = (int)((__float_as_uint(amax) >> 23) & 0xffu);
        int e = 13 - ex + 127;
        status = __builtin_amdgcn_readfirstlane(amax > 0.0f ? ((ex == 255 || e < -100) ? 2 : 0) : 1);
        e = amax > 0.0f ? (e < -100 ? -100 : (e > 113 ? 113 : e)) : 0;       // (2^(-e - 13) must stay a normal number)
        return __builtin_amdgcn_readfirstlane(e);
    };
    auto pass_scale = [&](int &status) { return scale_of(*reinterpret_cast<const floatx4 *>(red), status); };
    // where this thread's quad k lands in a staged buffer (halves): 4 tid + 1024 k, with SKEW + 4 ((4 tid + 1024 k) >> 7)
    const int sbase = SKEW == 1 ? 4 * tid + 4 * (tid >> 2) : (SKEW ? 4 * tid + 4 * ((4 * tid) / SKEW) : 4 * tid);
    constexpr int kinc = SKEW == 1 ? 5 * kBlock : (SKEW ? 4 * kBlock + 4 * ((4 * kBlock) / SKEW) : 4 * kBlock);

    // ---- prologue: pass 0 staged, pass 1 in the staging registers with its block maximum published
    int se_cur, se_m1 = 0;                // sample scale exponents of the pass in the matrix block and of the one before it
    int st_cur = 1;                       // the guard status of the pass in the matrix block (see scale_of)
    {
        const __amdgpu_buffer_rsrc_t rs = pass_rsrc(0);
#pragma unroll
        for (int k = 0; k < NL; k++) v0[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * tid + 16 * kBlock * k, 0, 0);
        const __amdgpu_buffer_rsrc_t rs1 = pass_rsrc(1);          // pass 1 follows at once: one exposed round trip, not two
#pragma unroll
        for (int k = 0; k < NL; k++) v1[k] = __builtin_amdgcn_raw_buffer_load_b128(rs1, 16 * tid + 16 * kBlock * k, 0, 0);
        max_partial(v0);
        __syncthreads();
        se_cur = pass_scale(st_cur);
        const float sx = pow2f(se_cur);
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const floatx4 q = as_floatx4(v0[k]);
            unsigned h0, l0, h1, l1;
            split_pair_scaled(q[0], q[1], sx, h0, l0);
            split_pair_scaled(q[2], q[3], sx, h1, l1);
            uint32x2 uh = {h0, h1}, ul = {l0, l1};
            _Float16 *ph = smp0 + sbase + kinc * k;
            *reinterpret_cast<uint32x2 *>(ph) = uh;
            *reinterpret_cast<uint32x2 *>(ph + d.r_smp_stride) = ul;
        }
        __syncthreads();                  // every wave has read the partial maxima of pass 0
        max_partial(v1);
        __syncthreads();
    }
    // diagnostic instantiation only (SYLDET_FUSED_STAMPS=1): s_memtime at the phase boundaries of every pass
    unsigned long long tsum[8] = {0}, tick[8] = {0};
    if (STAMP) {                          // slots 6, 7: the segment in shader clocks and in the fixed 100 MHz clock (their ratio: the clock held)
        tsum[6] = __builtin_amdgcn_s_memtime();
        tsum[7] = __builtin_amdgcn_s_memrealtime();
    }
#define SD_RTICK(slot)                                                                     \
    if (STAMP) {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        tick[slot] = __builtin_amdgcn_s_memtime();                                         \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    }
    SD_RTICK(5)

    // the evaluation's running values: this lane group's share of the window's products and sum of squares, the output, its flag
    float ssw = 1.0f, yv = 0.0f;
    bool hit = false;

    // ---- The three stages in flight.  In the matrix block of pass q this wave also finishes pass q-1 (magnitudes of
    // the accumulators it kept -> the column buffer of that parity, transition strip) and evaluates pass q-2 (from the
    // other column buffer): everything that crosses waves crosses a pass boundary, so one barrier per pass is enough,
    // and none of this work waits for the matrix pipe or has the matrix pipe wait for it.
    //
    // A lone wave issues in order: while it waits at an MFMA for the matrix pipe it issues nothing else, so vector work
    // hides under the matrix work only if it sits BETWEEN the MFMAs in program order (about three vector instructions
    // fit under each).  The block is therefore written as 12 KS "ticks", one per DFT MFMA, and every other piece of work
    // is cut into micro-steps assigned to ticks.  MFMAs are ordered assembly statements (also because the basis, only
    // ever an A operand, must live in the 256 accumulation registers, which the allocator will not do by itself); the
    // compiler keeps memory instructions on their side of such a statement, and vector instructions are held in place
    // by passing their inputs through empty ordered statements (pin) -- only where that matters, i.e. where a value comes
    // from LDS or memory and its wait must not be moved up: a lone wave is issue-bound, where a vector instruction sits is
    // immaterial, and a pin right behind a vector instruction costs a wait state.  Hazards the compiler cannot see:
    // accumulators are written by nothing else inside the block, consecutive uses of one accumulator are at least
    // three MFMAs apart, s_nops separate the last MFMA from vector reads of its result, and two wait states in front of
    // every MFMA cover a register-file move of an operand that the allocator may place right in front of it.
#define SD_PIN(x) asm volatile("" : "+v"(x))
    float amax_run = 0.0f, amax_b = 0.0f;
    floatx4 accP[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // pass q-1's DFT
    const int se_ref = se_cur;            // products are stored relative to the segment's first pass
    int r3 = 0;                           // q mod 3
    // -- precision guard (kernels.hpp, FixItem).  The window's sum of squares in the ring is relative to the segment's first
    // pass; the grid floor of a pass scaled by 2^se is phi 2^(se_ref - se) there.  An evaluation of pass p reads columns of
    // passes p-1 and p: its threshold comes from the louder of the two (a silent pass holds exact zeros and does not count;
    // a pass the grid cannot hold, or one 2^45 away from the segment's first, condemns every window that touches it).
    // Wave-uniform integer arithmetic: a power of two times the host's constant is an addition to its exponent field.
    const int gnorm = GEN ? d.norm : 1;
    auto guard_thr = [&](int se_a, int st_a, int se_b, int st_b) -> unsigned {
        if (st_a == 2 || st_b == 2) return 0x7f800000u;                   // +inf: nothing passes
        if (st_a == 1 && st_b == 1) return 0u;                            // exact zeros: the fused result is the reference's (0/0)
        const int m = st_a == 1 ? se_b : (st_b == 1 ? se_a : (se_a < se_b ? se_a : se_b));
        const int dm = se_ref - m;
        if (dm > 45 || dm < -45) return 0x7f800000u;
        if (gnorm != 1 && m >= d.guard_se_abs_r) return 0u;               // no normaliser: too quiet for the floor to matter
        return __float_as_uint(gnorm == 1 ? d.guard_r : d.guard_rel_r) + ((unsigned)dm << 24);
    };
    unsigned thr_0 = guard_thr(0, 1, se_cur, st_cur), thr_m1 = 0u, thr_m2 = 0u;   // evaluations of passes q, q-1, q-2
    bool badv = false;                    // this lane's evaluation of the pass failed the guard
    const bool guard_on = d.fix.counters != nullptr;

    // -- evaluation of pass pp (ring region re): every column met every tap when its pass was finished, so an evaluation
    // is the diagonal sum over the taps of the products in LDS (lane group g4 takes taps g4, g4 + 4, g4 + 8; taps past
    // timeRange are rows of zeros) plus the window's sum of squares, then the rest of the network.  Slots: one every
    // four ticks; the fetches two slots ahead of the sums.
    floatx4 pv[3], zp = {0.f, 0.f, 0.f, 0.f};
    float sv[3], ssp = 0.0f;
    // The sums over the four lane groups, as a halving butterfly: v_permlane16_swap of (z0, z1) leaves the pair sum of z0 in
    // the even 16-lane rows and that of z1 in the odd ones, likewise (z2, z3); v_permlane32_swap of the two results leaves
    // the total of z[g4] in lane group g4 -- three swaps and three adds for the four units, and each lane group then runs
    // ONE hidden unit (its own) through the transfer function instead of all four.  One step a tick (ticks 9 .. 12).
    float s01 = 0.0f, s23 = 0.0f, zt = 0.0f;
    const float b0g = g4 < H ? d.bias0[g4] : 0.0f, w1g = g4 < H ? d.w1[g4] : 0.0f;   // this lane group's hidden unit
    float w1o[4];                                     // (GEN, several outputs: its weight in every output, w1 is [output][unit])
#pragma unroll
    for (int o = 0; o < 4; o++) w1o[o] = (GEN && g4 < H && o < n_out) ? d.w1[o * H + g4] : 0.0f;
    const bool multi = GEN && n_out > 1;
    const double thr_g = d.thresholds[(GEN && g4 < n_out) ? g4 : 0];
    const bool counts = !GEN || n_out == 1 ? true : (g4 < n_out && (g4 == 0 || d.rule == 1));    // lastDetected: output 0; CLI rule: any
    auto eval_reduce = [&](int k) {
        if (k == 0) {
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(zp[0]), __float_as_uint(zp[1]), false, false);
            s01 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        } else if (k == 1) {
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(zp[2]), __float_as_uint(zp[3]), false, false);
            s23 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        } else if (k == 2) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
            zt = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        } else if (k == 3) {
            if (GEN && gnorm == 0) {                              // (the guard's statistic without a normaliser: the quietest column)
                auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(ssp), __float_as_uint(ssp), false, false);
                const float m = fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
                r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                ssw = fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
            } else {
                ssw = xor32_sum(xor16_sum(ssp));
            }
        }
    };
    // the rest of the network (NeuralNet.swift:47-59 L2Normalize on the folded first layer, :189-194 TanSig, :366-377 second
    // layer, :137-142 / :175-180 reverse output map; SyllableDetector.swift:27-31 threshold), one step a slot
    float ypart = 0.0f, yp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int seg_len = (int)(e_e - e_b);
    const unsigned e_b32 = (unsigned)e_b;
    // (GEN: without a normaliser the products, kept relative to the segment's first pass, go back to true units by a constant)
    const int tf0 = GEN ? d.tf0 : 0, tf1 = GEN ? d.tf1 : 2, norm = GEN ? d.norm : 1;
    const int ush = d.col_shift - se_ref;
    const float alpha0 = d.w_unscale * pow2f(ush < -120 ? -120 : (ush > 120 ? 120 : ush));
    auto eval_tail = [&](int k, int pp, unsigned thr) {
        if (k == 0) {                                             // this group's unit: z and the sums of squares are both relative
            const float alpha = norm == 1 ? d.w_unscale * __builtin_amdgcn_rsqf(ssw) : alpha0;   // L2Normalize, NeuralNet.swift:47-59
            const float act = transfer_fn(tf0, fmaf(alpha, zt, b0g));
            ypart = w1g * act;
            if (multi) {
#pragma unroll
                for (int o = 0; o < 4; o++) yp[o] = w1o[o] * act;
            }
        } else if (k == 1) {
            float ysum;
            if (multi) {
                // the halving butterfly of eval_reduce again: the sum over the lane groups (hidden units) of yp[o] lands in lane
                // group o, which finishes output o
                auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(yp[0]), __float_as_uint(yp[1]), false, false);
                const float a01 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                r = __builtin_amdgcn_permlane16_swap(__float_as_uint(yp[2]), __float_as_uint(yp[3]), false, false);
                const float a23 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a01), __float_as_uint(a23), false, false);
                ysum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            } else {
                ysum = xor32_sum(xor16_sum(ypart));
            }
            float y = transfer_fn(tf1, ysum + c_b1);
            y = (y - lean_oa) / lean_og + lean_ob;
            yv = y;
            hit = counts && (double)y >= thr_g;
            if (multi) {                                          // one flag an evaluation: any counting output over its threshold
                unsigned hb = hit ? 1u : 0u;
                auto r = __builtin_amdgcn_permlane16_swap(hb, hb, false, false);
                hb = r[0] | r[1];
                r = __builtin_amdgcn_permlane32_swap(hb, hb, false, false);
                hit = (r[0] | r[1]) != 0u;
            }
        } else if (k == 2) {                                      // stores, through the bounds-checked descriptors of this channel's rows
            const int er = kPass * pp - (T - 1) + fl;             // evaluation index inside the segment (32-bit arithmetic)
            const bool vld = pp >= 0 && er >= 0 && er < seg_len;
            const bool st = vld && g4 == 0, sto = vld && g4 < n_out;      // flags: lane group 0; outputs: a lane group each
            badv = vld && !(ssw >= __uint_as_float(thr));         // the guard: too close to the grid's floor (or NaN) -> work list
            const unsigned off = e_b32 + (unsigned)er;            // E * 4 < 2^32 is checked by the launcher
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yv), out_rs, sto ? (off * (unsigned)n_out + (unsigned)g4) * 4u : 0xFFFFFFFFu, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(hit ? 1 : 0), flg_rs, st ? off : 0xFFFFFFFFu, 0, 0);
        }
    };
    const float *zero_ss = pbuf + (kPRows - 1) * kPStride + 48;       // floats 36 .. 48 of the last row stay zero
    auto eval_slot = [&](int s, int pp, int re, unsigned thr) {
        if (s == 0) {
            const float *erow = pbuf + (kPLead + 64 * re + fl - (T - 1)) * kPStride;
#pragma unroll
            for (int tt = 0; tt < 3; tt++) {
                const int t = g4 + 4 * tt;
                // (taps past timeRange: their products are 0 * column, which is NaN for a column with a NaN in it -- read zeros)
                pv[tt] = *reinterpret_cast<const floatx4 *>(t < T ? erow + t * kPStride + 4 * t : zero_ss - 12);
                sv[tt] = *(t < T ? erow + t * kPStride + 48 : zero_ss);
            }
        }
        if (s == 2) {                                             // this lane group's taps
#pragma unroll
            for (int tt = 0; tt < 3; tt++) {
                SD_PIN(pv[tt]);
                SD_PIN(sv[tt]);
            }
            zp = pv[0] + pv[1] + pv[2];
            if (GEN && gnorm == 0)                                // no normaliser: the guard wants the quietest column of the window
                ssp = fminf(fminf(g4 < T ? sv[0] : INFINITY, g4 + 4 < T ? sv[1] : INFINITY), g4 + 8 < T ? sv[2] : INFINITY);
            else
                ssp = sv[0] + sv[1] + sv[2];
        }
        if (s >= 4 && s <= 8 && s % 2 == 0) eval_tail(s / 2 - 2, pp, thr);
    };
    // evaluations of pass pp that failed the guard: this wave's 16 go to the work list as one item (rare: a branch outside
    // the matrix block)
    auto push_bad = [&](int pp) {
        if (guard_on && __builtin_amdgcn_ballot_w64(badv) != 0ull) {
            const int er0 = kPass * pp - (T - 1) + 16 * wave;
            const int lo = er0 < 0 ? 0 : er0, hi = er0 + 16 < seg_len ? er0 + 16 : seg_len;
            if (lane == 0 && hi > lo) {
                const unsigned slot = atomicAdd(d.fix.counters, 1u);
                if (slot < d.fix.capacity) d.fix.items[slot] = FixItem{c, e_b32 + (unsigned)lo, hi - lo, 0};
                else d.fix.counters[3] = 1u;
            }
        }
        badv = false;
    };

    // -- finishing pass q-1: magnitudes (zvabs/2, CircularShortTimeFourierTransform.swift:329-333), their f16 hi + lo split,
    // the frame's sum of squares, and the tap products of the first layer.  Result layout of the DFT: column = frame f,
    // register j of lane group g4 in tile m = basis row 16m + 4*g4 + j; this lane holds bins 4*g4 + j (i = j) and
    // 16 + 4*g4 + j (i = 4 + j) -- which IS the B operand of an MFMA (lane (f, g4): column f, k = 8 g4 + i) for a first
    // layer whose columns are permuted accordingly on the host (afrag_t): no LDS round trip for the columns.  accP holds
    // X * 2^(se + 13); |X| * 2^(se - col_shift) is what is split; products and sums of squares are stored relative to the
    // segment's first pass (* 2^dsc, dsc = se(first) - se: a power of two), so that a window may straddle passes of
    // different scales.  Micro-steps j = 0 .. kMagSteps-1.
    constexpr int kMagSteps = 27;
    float cval[8], mss = 0.0f, fs_up = 1.0f, fs_ring = 1.0f;
    unsigned bh[4], bl[4];
    floatx4 pt[3];
    const float kmag = pow2f(-13 - d.col_shift);
    // (inblock: called from the matrix block, where DFT MFMAs and other work sit between the steps; the drain calls them back to back)
    auto mag_micro = [&](int j, int rm, int dsc, bool inblock) {
        // this frame's row in the ring; the ring's last T-1 frames are repeated in front of it (elsewhere: a spare spot)
        float *prow = pbuf + (kPLead + 64 * rm + fl) * kPStride;
        // (only frames of the pass's last wave, in every third pass: those stores sit behind a test -- a wave has one LDS
        // write in flight at a time, 55 clocks for a b128, so a store that is not needed is not free)
        const bool dupl = rm == 2 && fl >= kPass - (T - 1);
        float *drow = pbuf + (kPLead + fl - kPass) * kPStride;
        float *spare = pbuf + (kPRows - 1) * kPStride;             // floats 0 .. 35 and 49 .. of the last row: where lanes with nothing to store store
        if (j < 8) {                                              // |X| of one bin
            const int i = j;
            float re = accP[i >> 2][i & 3], im = accP[2 + (i >> 2)][i & 3];
            cval[i] = __builtin_amdgcn_sqrtf(fmaf(re, re, im * im)) * kmag;
        } else if (j < 12) {                                      // the frame's sum of squares (l2normalize works from these)
            const int i = 2 * (j - 8);
            if (j == 8) mss = 0.0f;
            mss = fmaf(cval[i], cval[i], mss);
            mss = fmaf(cval[i + 1], cval[i + 1], mss);
            if (j == 11) mss = xor32_sum(xor16_sum(mss));
        } else if (j == 12) {
            const float sr = mss * pow2f(2 * dsc);
            *(g4 == 0 ? prow + 48 : spare + 49) = sr;
            if (dupl && g4 == 0) drow[48] = sr;
            // The frame's own column exponent: its column is split into f16 hi + lo at the scale that puts its norm into
            // [2^12, 2^13) -- a quiet frame of a loud pass keeps 22 bits of its own level instead of the pass's f16 floor --
            // and the products come back through fs_ring = 2^dsc / fs_up.  With ex the biased exponent of mss,
            // t = floor((ex + 1) / 2) = floor(log2 sqrt(mss)) + 64, clamped to [16, 80]: fs_up = 2^(76 - t).
            unsigned tb = ((__float_as_uint(mss) + 0x800000u) >> 1) & 0x7f800000u;    // t << 23
            tb = (unsigned)min(max((int)tb, 16 << 23), 80 << 23);                    // v_med3_i32 (NaN / inf: 80; zero: 16)
            fs_up = __uint_as_float((203u << 23) - tb);
            fs_ring = __uint_as_float(tb + ((unsigned)(dsc + 51) << 23));
        } else if (j < 15) {                                      // f16 hi + lo of four bins: half of the B operand pair (the
            const int m = j - 13;                                 // staging's split: one-slot instructions, the same bits)
            float ta, tb2, tc, td, ra, rb, rc, rd;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ta) : "v"(cval[4 * m]), "v"(fs_up));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(tb2) : "v"(cval[4 * m + 1]), "v"(fs_up));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(tc) : "v"(cval[4 * m + 2]), "v"(fs_up));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(td) : "v"(cval[4 * m + 3]), "v"(fs_up));
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(bh[2 * m]) : "v"(ta), "v"(tb2));
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(bh[2 * m + 1]) : "v"(tc), "v"(td));
            asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(cval[4 * m]), "v"(fs_up), "v"(bh[2 * m]));
            asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(rc) : "v"(cval[4 * m + 2]), "v"(fs_up), "v"(bh[2 * m + 1]));
            asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(cval[4 * m + 1]), "v"(fs_up), "v"(bh[2 * m]));
            asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rd) : "v"(cval[4 * m + 3]), "v"(fs_up), "v"(bh[2 * m + 1]));
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(bl[2 * m]) : "v"(ra), "v"(rb));
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(bl[2 * m + 1]) : "v"(rc), "v"(rd));
        } else if (j >= 16 && j < 19) {                           // tap products: hi*hi, hi*lo, lo*hi, one term of every row tile a step
            const uint32x4 vbh = {bh[0], bh[1], bh[2], bh[3]}, vbl = {bl[0], bl[1], bl[2], bl[3]};
#pragma unroll
            for (int m = 0; m < 3; m++) {
                if (j == 16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(pt[m]) : "v"(aft[m][0]), "v"(vbh));
                if (j == 17) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(pt[m]) : "v"(aft[m][0]), "v"(vbl));
                if (j == 18) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(pt[m]) : "v"(aft[m][1]), "v"(vbh));
            }
        } else if (j >= 20 && j < 23) {                           // tile m: taps 4m + g4, units 0..3 of this frame -> its row
            const int m = j - 20;
            // (eight wait states behind the last tap MFMA; in the block two ticks -- two DFT MFMAs, a load, staging steps --
            // lie between, the drain has nothing there)
            if (m == 0) {
                if (inblock) asm volatile("s_nop 3" : "+v"(pt[0]), "+v"(pt[1]), "+v"(pt[2]));
                else asm volatile("s_nop 7" : "+v"(pt[0]), "+v"(pt[1]), "+v"(pt[2]));
            }
            pt[m] = pt[m] * fs_ring;
            *reinterpret_cast<floatx4 *>(prow + 4 * (4 * m + g4)) = pt[m];
        } else if (j >= 24) {                                     // the ring's last T-1 frames once more in front of it
            const int m = j - 24;
            if (dupl) *reinterpret_cast<floatx4 *>(drow + 4 * (4 * m + g4)) = pt[m];
        }
    };

    // pass q+2's samples: where they start, how many are left from there to the end of the stream (clamped into 32 bits once)
    const float *base2 = row + (e_b + (int64_t)kPass * 2) * d.hop + d.gap;
    int left2;
    {
        int64_t l = s_eff - ((e_b + (int64_t)kPass * 2) * d.hop + d.gap);
        l = l < -(int64_t)0x3fffffff ? -(int64_t)0x3fffffff : (l > 0x3fffffff ? 0x3fffffff : l);
        left2 = __builtin_amdgcn_readfirstlane((int)l);
    }
    // One pass, with its parity a compile-time fact (which staging set is staged from, which is loaded into).
    auto pass_body = [&](auto par_tag, const int q) {
        constexpr int par = decltype(par_tag)::value;
        uint32x4 (&vs)[NL] = par ? v0 : v1;                           // pass q+1's samples: staged in this block
        uint32x4 (&vl)[NL] = par ? v1 : v0;                           // pass q+2's samples: loaded in this block, maximum taken at its end
        const _Float16 *fph = smp0 + par * buf_halves + foff, *fpl = fph + d.r_smp_stride;
        half8 bh = lds_half8(fph + ko[0]), bl = lds_half8(fpl + ko[0]);   // first B fragments: all the first MFMA waits for
        floatx4 r0 = *reinterpret_cast<const floatx4 *>(red);             // pass q+1's partial maxima
        // In the shadow of those fetches: the vector half of finishing pass q-1 (magnitudes, sum of squares, f16 split; it
        // needs nothing but the accumulators this wave kept), which would otherwise crowd the second half of the block.
        const int dsc = se_ref - se_m1 < -45 ? -45 : (se_ref - se_m1 > 45 ? 45 : se_ref - se_m1);   // (beyond: the guard's business)
        const int rm = r3 == 0 ? 2 : r3 - 1, re = r3 == 2 ? 0 : r3 + 1;   // (q - 1) mod 3, (q - 2) mod 3
#if !defined(SYLDET_R_NOMAG) && !defined(SYLDET_R_MAGTICK)
#pragma unroll
        for (int j = 0; j < 15; j++) mag_micro(j, rm, dsc, true);
#endif
        SD_PIN(r0);
        int st_next;
        const int se_next = scale_of(r0, st_next);                    // pass q+1's sample scale
        const float sx_next = pow2f(se_next);
        _Float16 *wh = smp0 + (par ^ 1) * buf_halves;
        // pass q+2's descriptor, kept incrementally in 32-bit scalar arithmetic (pass_rsrc's 64-bit clamps cost thirty
        // instructions a pass): samples left from its first one, none past the segment
        int left = left2 < 0 ? 0 : (left2 > d.r_nsmp ? d.r_nsmp : left2);
        left = q + 2 < runs ? left : 0;
#ifdef SYLDET_R_NOLOAD                // (diagnostic builds reload the same cache-resident pass: tools/r_knockouts.sh)
        const __amdgpu_buffer_rsrc_t rs2 = pass_rsrc(q & 1);
#else
        const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base2), 0, left * 4, 0x00020000);
#endif
        base2 += kPass * d.hop;
        left2 -= kPass * d.hop;
        // (pass q-1's products go to ring region rm relative to the segment's first pass; pass q-2's are read from region re)
        floatx4 acc[4];
        {
            // the next pass's staging (scale, f16 hi/lo split, two LDS writes), one instruction per micro-step, one
            // micro-step a tick: v_fma_mix runs at half rate, one rides under an MFMA and a second one does not (tools/ubench)
            // The split, priced next to an MFMA on a lone wave (tools/ubench/tick_costs): two plain vector instructions ride
            // free behind every MFMA, a third costs an issue slot; v_fma_mixlo/hi_f16 takes two slots, a packed fp32
            // instruction waits for the matrix pipe (+17 clocks), and a wave has one LDS write in flight per ~32 clocks (b64;
            // ~55 for b128).  Hence: v_mul_f32, v_cvt_pk_f16_f32, v_fma_mix_f32 (one slot each; the same bits -- x*s is
            // exact, s being a power of two, hi = RNE(x*s), x*s - hi is exact in fp32, lo = RNE of it), 12 slots a quad instead
            // of 16, and the quad's two LDS writes seven steps apart.
            unsigned mh0 = 0, ml0 = 0, mh1 = 0, ml1 = 0;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
            constexpr int kStageSteps = 14;
            auto stage_micro = [&](int i) {
                const int k = i / kStageSteps, j = i % kStageSteps;
                if (k >= NL) return;
                const floatx4 qv = as_floatx4(vs[k]);
                _Float16 *ph = wh + sbase + kinc * k;
                if (j == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(qv[0]), "v"(sx_next));
                if (j == 1) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(qv[1]), "v"(sx_next));
                if (j == 2) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t2) : "v"(qv[2]), "v"(sx_next));
                if (j == 3) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t3) : "v"(qv[3]), "v"(sx_next));
                if (j == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(mh0) : "v"(t0), "v"(t1));
                if (j == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(mh1) : "v"(t2), "v"(t3));
                if (j == 6) { uint32x2 uh = {mh0, mh1}; *reinterpret_cast<uint32x2 *>(ph) = uh; }
                if (j == 7) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(qv[0]), "v"(sx_next), "v"(mh0));
                if (j == 8) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l2) : "v"(qv[2]), "v"(sx_next), "v"(mh1));
                if (j == 9) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(qv[1]), "v"(sx_next), "v"(mh0));
                if (j == 10) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l3) : "v"(qv[3]), "v"(sx_next), "v"(mh1));
                if (j == 11) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ml0) : "v"(l0), "v"(l1));
                if (j == 12) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ml1) : "v"(l2), "v"(l3));
                if (j == 13) { uint32x2 ul = {ml0, ml1}; *reinterpret_cast<uint32x2 *>(ph + d.r_smp_stride) = ul; }
            };
            // tick i: what rides behind the i-th DFT MFMA.  The loads of pass q+2 leave in the first ticks (the other staging
            // set is free) and have the whole block to land before the block maximum reads them in the last ticks; staging
            // one micro-step a tick throughout; the evaluation of pass q-2 in the first half, the finishing of pass q-1
            // behind it.  (Measured before the second staging set existed: two staging steps a tick in the first half with
            // each quad reloaded as soon as it was staged, 1.37 ms; one a tick with the block maximum behind the block, 1.42.)
            constexpr int kTicks = 12 * KS;
            auto tick_work = [&](int i) {
                if (STAMP && i % 24 == 0) { SD_RTICK(i / 24) }         // quarters of the block, diagnostic instantiation only
                // (SYLDET_R_NO*: diagnostic builds with one piece knocked out, tools/r_knockouts.sh; never the shipped library)
                if (i < NL) vl[i] = __builtin_amdgcn_raw_buffer_load_b128(rs2, 16 * tid + 16 * kBlock * i, 0, 0);
#ifndef SYLDET_R_NOSTAGE
                // 14 NL = 126 micro-steps over the 96 ticks: 21 every 16 ticks
#pragma unroll
                for (int sm = i * (kStageSteps * NL) / kTicks; sm < (i + 1) * (kStageSteps * NL) / kTicks; sm++) stage_micro(sm);
#endif
#ifndef SYLDET_R_NOEVAL
                if (i % 4 == 0) eval_slot(i / 4, q - 2, re, thr_m2);
                if (i >= 9 && i < 13) eval_reduce(i - 9);
#endif
#ifndef SYLDET_R_NOMAG
#ifdef SYLDET_R_MAGTICK                // (experiment: the whole finishing of pass q-1 inside the block, from this tick on)
                const int jm = i - SYLDET_R_MAGTICK;
                if (jm >= 0 && jm < kMagSteps) mag_micro(jm, rm, dsc, true);
#else
                // pass q-1's tap products in ticks 2-4, their stores four ticks apart (one b128 LDS write of a wave in
                // flight at a time), the copies in front of the ring behind them
                if (i >= 2 && i <= 4) mag_micro(i + 14, rm, dsc, true);
                if (i == 6 || i == 10 || i == 14) mag_micro(20 + (i - 6) / 4, rm, dsc, true);
                if (i == 18 || i == 22 || i == 26) mag_micro(24 + (i - 18) / 4, rm, dsc, true);
#endif
#endif
#ifndef SYLDET_R_NOMAX
                const int jx = i - (kTicks - NL - 4);                  // block maximum of pass q+2, a quad a tick (its loads left in the
                if (jx >= 0 && jx < NL) {                              // first half), then the wave's maximum, published before the last tick
                    // (the quad passes through an ordered statement HERE, so that the wait for its load is not moved up; two
                    // running maxima, so that no v_max3 reads the one in front of it)
                    if (jx == 0) { amax_run = 0.0f; amax_b = 0.0f; }
                    SD_PIN(vl[jx]);
                    const floatx4 qv = as_floatx4(vl[jx]);
                    amax_run = absmax3(amax_run, qv[0], qv[1]);
                    amax_b = absmax3(amax_b, qv[2], qv[3]);
                }
                if (jx == NL) amax_run = wave_max_nonneg(fmaxf(amax_run, amax_b));   // v_max3 drops NaNs: plain non-negative numbers
                if (jx == NL + 2 && lane == 0) red[wave] = amax_run;
#endif
            };
            static_assert(NL <= 9 && (KS == 8 || KS == 4) && TMAX <= 12 && TMAX - 1 <= kPLead, "tick schedule");
            // The DFT's accumulators live in the accumulation registers (an MFMA whose C/D operands are architectural
            // registers takes 9.9 ns against 8.4: tools/ubench), which the basis alone would fill: its last four quads are
            // architectural instead.
            constexpr int kArchQuads = 9;
#define SD_DFT_MFMA(m_, ai_, b_, first_)                                                                                           \
            if ((ai_) >= KS * 8 - kArchQuads) {                                                                                   \
                if (first_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&a"(acc[m_]) : "v"(a[ai_]), "v"(b_));         \
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[m_]) : "v"(a[ai_]), "v"(b_));                \
            } else {                                                                                                              \
                if (first_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&a"(acc[m_]) : "a"(a[ai_]), "v"(b_));         \
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[m_]) : "a"(a[ai_]), "v"(b_));                \
            }
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const half8 cbh = bh, cbl = bl;
                if (ks + 1 < KS) {                                    // B fragments of the next k-step: a whole k-step ahead
                    bh = lds_half8(fph + ko[ks + 1]);
                    bl = lds_half8(fpl + ko[ks + 1]);
                }
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    SD_DFT_MFMA(m, ks * 8 + 2 * m, cbh, ks == 0)
                    tick_work(12 * ks + m);
                }
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    SD_DFT_MFMA(m, ks * 8 + 2 * m, cbl, false)
                    tick_work(12 * ks + 4 + m);
                }
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    SD_DFT_MFMA(m, ks * 8 + 2 * m + 1, cbh, false)
                    tick_work(12 * ks + 8 + m);
                }
            }
#undef SD_DFT_MFMA
            // (eight wait states between an MFMA of this shape and a vector read of its result: what the compiler places there)
            asm volatile("s_nop 7" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
#pragma unroll
            for (int m = 0; m < 4; m++) accP[m] = acc[m];
        }
        SD_RTICK(4)
        thr_m2 = thr_m1;
        thr_m1 = thr_0;
        thr_0 = guard_thr(se_cur, st_cur, se_next, st_next);
        se_m1 = se_cur;
        se_cur = se_next;
        st_cur = st_next;
        r3 = r3 == 2 ? 0 : r3 + 1;
        __syncthreads();          // products of pass q-1, staged samples of pass q+1 and the partial maxima are complete
        push_bad(q - 2);
        SD_RTICK(6)
        if (STAMP) {                // top of the pass up to the first MFMA, four quarters of the block, barrier
            tsum[0] += tick[0] - tick[5];
#pragma unroll
            for (int i = 1; i < 5; i++) tsum[i] += tick[i] - tick[i - 1];
            tsum[5] += tick[6] - tick[4];
            tick[5] = tick[6];
        }
    };
    for (int q = 0; q < runs; q += 2) {
        pass_body(std::integral_constant<int, 0>{}, q);
        if (q + 1 < runs) pass_body(std::integral_constant<int, 1>{}, q + 1);
    }
    // ---- drain: evaluate pass runs-2, finish pass runs-1, barrier, evaluate it
    {
        const int q = runs;
        const int dsc = se_ref - se_m1 < -45 ? -45 : (se_ref - se_m1 > 45 ? 45 : se_ref - se_m1);   // (beyond: the guard's business)
        const int rm = r3 == 0 ? 2 : r3 - 1, re = r3 == 2 ? 0 : r3 + 1;
#pragma unroll
        for (int sl = 0; sl < 12; sl++) {
            eval_slot(sl, q - 2, re, thr_m2);
            if (sl == 2)
                for (int k = 0; k < 4; k++) eval_reduce(k);
        }
        push_bad(q - 2);
#pragma unroll
        for (int j = 0; j < kMagSteps; j++) mag_micro(j, rm, dsc, false);
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < 12; sl++) {
            eval_slot(sl, q - 1, rm, thr_m1);
            if (sl == 2)
                for (int k = 0; k < 4; k++) eval_reduce(k);
        }
        push_bad(q - 1);
    }
    if (STAMP) {
        tsum[6] = __builtin_amdgcn_s_memtime() - tsum[6];
        tsum[7] = __builtin_amdgcn_s_memrealtime() - tsum[7];
    }
    if (STAMP && (tid == 0 || tid == 64 * (kWaves - 1)) && d.stamps)       // wave 0's view in slots 0-7, the last wave's in 8-15
        for (int i = 0; i < 8; i++) atomicAdd(&d.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (tid ? 8 : 0) + i], tsum[i]);
#undef SD_RTICK
#undef SD_PIN
}

template <int KS, int TMAX, int NL, int SKEW, bool STAMP = false, bool GEN = false>
hipError_t launch_one(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t s_eff, int64_t E,
                      float *outputs, uint8_t *flags, hipStream_t stream)
{
    auto kern = fused_r_kernel<KS, TMAX, NL, SKEW, STAMP, GEN>;
    hipError_t st = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, d.r_lds_total);
    if (st != hipSuccess) return st;
    const int64_t segs = (E + d.r_seg_evals - 1) / d.r_seg_evals;
    dim3 grid((unsigned)segs, (unsigned)C);
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), (size_t)d.r_lds_total, stream, d, samples, stride, s_eff, E, outputs, flags);
    return hipGetLastError();
}

}  // namespace

// Shapes this kernel is instantiated for: the reference's example detector class (l2normalize first, linear |X| columns,
// two layers, at most 4 TanSig hidden units, one output, at most one output map), windows of 132 .. 256 samples (8
// k-steps) or up to 128 (4), timeRange up to 12, at most 9 staging quads per thread (hop <= 140, the reference's 132 among them), no
// bank-spreading padding except at hops 16, 32, 64 and 128 (a table-driven instantiation for any multiple of 16 spilled and
// measured 1.85 ms against the 8-wave kernel's 1.47; at a power of two the padding is a matter of constants, and the three
// short hops -- 75 % overlap and more -- stage 2, 3 and 5 quads a thread instead of 9).  Everything else stays on
// kernels_fused.hip's kernel.
bool fused_r_has_stamps()
{
#ifdef SYLDET_R_STAMPS
    return true;
#else
    return false;
#endif
}

bool fused_r_applicable(const FusedDesc &d)
{
    // two layers with at most 4 hidden units and one output, linear |X| columns, no normaliser or l2normalize in front of the
    // affine maps, at most one output map (any transfer functions; the example detector's get the exact instantiation)
    const bool cls = (d.norm == 0 || d.norm == 1) && d.scaling == 0 && d.n_layers == 2 && d.n_out >= 1 && d.n_out <= 4 && d.H <= 4 && d.n_out_fns <= 1;
    return d.r_ok && (d.KS == 8 || d.KS == 4) && d.T <= 12 && d.r_nload <= 9 && (d.skew == 0 || d.skew == 4) && cls;
}

hipError_t launch_fused_r(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t S, int64_t J,
                          int64_t E, float *outputs, uint8_t *flags, hipStream_t stream)
{
    (void)S;
    if (E <= 0 || C <= 0) return hipSuccess;
    if (!fused_r_applicable(d)) return hipErrorInvalidValue;
    const int64_t s_eff = (J - 1) * (int64_t)d.hop + d.gap + d.W;
#ifdef SYLDET_R_STAMPS                // diagnostic builds only (-DSYLDET_R_STAMPS): the instantiation with phase timing
    if (d.stamps && d.skew == 0 && d.KS == 8) return launch_one<8, 12, 9, 0, true>(d, samples, stride, C, s_eff, E, outputs, flags, stream);
#endif
    const bool exact = d.norm == 1 && d.tf0 == 0 /* TanSig */ && d.tf1 == 2 /* PureLin */ && d.n_out == 1;     // the example detector's class
    // (the short hops, and windows of up to 128 samples -- four k-steps --, take the instantiation with the network class as
    // run-time facts for the example class too)
#define SD_R_GO(KS_, NL_, SKEW_, GEN_) return launch_one<KS_, 12, NL_, SKEW_, false, GEN_>(d, samples, stride, C, s_eff, E, outputs, flags, stream)
    const bool pad16 = d.skew != 0 && d.hop != 16 && d.hop != 32 && d.hop != 64 && d.hop != 128;      // 48, 80, 96, 112
    if (d.KS == 4) {
        if (pad16) { if (d.r_nload <= 6) SD_R_GO(4, 6, 1, true); SD_R_GO(4, 9, 1, true); }
        if (d.skew != 0 && d.hop == 16) SD_R_GO(4, 2, 16, true);
        if (d.skew != 0 && d.hop == 32) SD_R_GO(4, 3, 32, true);
        if (d.skew != 0 && d.hop == 64) SD_R_GO(4, 5, 64, true);
        if (d.skew != 0) SD_R_GO(4, 9, 128, true);
        SD_R_GO(4, 9, 0, true);
    }
    if (pad16) { if (d.r_nload <= 6) SD_R_GO(8, 6, 1, true); SD_R_GO(8, 9, 1, true); }
    if (d.skew != 0 && d.hop == 16) SD_R_GO(8, 2, 16, true);
    if (d.skew != 0 && d.hop == 32) SD_R_GO(8, 3, 32, true);
    if (d.skew != 0 && d.hop == 64) SD_R_GO(8, 5, 64, true);
    if (d.skew != 0) { if (exact) SD_R_GO(8, 9, 128, false); SD_R_GO(8, 9, 128, true); }
    if (exact) SD_R_GO(8, 9, 0, false);
    SD_R_GO(8, 9, 0, true);
#undef SD_R_GO
}

}  // namespace sd
