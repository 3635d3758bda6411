// kernels_fft1k.hip -- 1024-point frames (BASELINE configs[2]: hop 256, 116 bins, 512 channels) in ONE kernel: samples in
// HBM -> network outputs + detection flags in HBM, the spectrogram never leaves the CU.
//
// Reference path being replaced, per frame and per evaluation (reference root relative):
//   extractPower          Common/CircularShortTimeFourierTransform.swift:280-337   (window, packed real FFT, |X|)
//   processFourierData    Common/SyllableDetector.swift:134-151                    (slice to [f0, f1))
//   processNewValue       Common/SyllableDetector.swift:153-217                    (timeRange-column window)
//   NeuralNet.apply       Common/NeuralNet.swift:294-326, :366-377                 (l2normalize, affine maps, TanSig, linear)
//   lastDetected          Common/SyllableDetector.swift:27-31
//
// The two halves are the ones the generic engine runs as two launches with the [C][J][F] columns in HBM between them
// (stft_r8_kernel, kernels_generic.hip: 464 B written + 464 B re-read per frame against 1029 algorithmic bytes):
//   * the packed real FFT of a frame as 512 = 8 x 8 x 8 complex points, one wave per frame, eight points per lane, three
//     radix-8 passes in registers with two transposes through a wave-private 4.5 KB of LDS, real split and |X| for the band;
//   * the first layer with ALL taps as the rows of one GEMM on the matrix cores (kernels_mlpx.hip's formulation:
//     P[(t, h), j] = W'_t[h, :] . c(j), f16 hi + lo operands, three products, fp32 accumulate), evaluations as diagonal sums.
// Here a workgroup (12 waves) walks a contiguous run of 128-frame tiles of one channel: its waves transform the tile's new
// frames (10 or 11 each), every frame's |X| column goes to LDS as f16 hi + lo under the frame's OWN power-of-two exponent (a quiet
// frame next to a loud one keeps 22 bits of its own level), the tile's tap products come off the matrix cores, one thread per
// evaluation finishes the network, and the last timeRange - 1 columns are carried to the front of the next tile.
//
// HBM traffic: every sample once (+ the W - hop overlap of consecutive frames, L2 hits) + 5 bytes per evaluation.
// Taken by AUTO when the window is 1024 samples without zero padding, frames start 8-byte aligned, and the network is of
// the matrix-core class (make_mlpx_plan); everything else keeps the two-launch path.
//
// gfx950 only.  wave = 64.

#include "fused_common.hpp"

namespace sd {

namespace {

using namespace fused_dev;

constexpr int kBlock = kFft1kBlock;            // 768 threads = 12 waves: the transform is a chain of LDS round trips, and three
                                               // waves per SIMD hide more of it than two (3.08 ms against 3.28 on the BASELINE
                                               // configs[2] batch; four, with the first layer's fragments read from memory to
                                               // make room for their scratch, spill: 3.82 ms)
constexpr int kWaves = kBlock / 64;
constexpr int kTile = 128;                     // frames per tile
constexpr int kScratch = 8 * 80;               // float2 per frame in flight: rows of 64 (+8) / 8 x 8 rows of 8 (+1) / 512 in natural order, rows of 64 skewed to 80
// Frames a wave transforms at a time.  Measured (BASELINE configs[2] batch): two frames in flight on 64-frame tiles (the LDS
// budget of a second scratch set) 4.5 ms against 3.4 ms -- the transform is bound by LDS throughput (26 KB of transposes per
// frame, stores at ~80 B/clk), not by the latency a second chain would hide.
constexpr int kFly = 1;

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 cmul(f2 a, f2 b) { return f2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// 8-point DFT in place, outputs in natural order (radix-2 decimation in frequency; see kernels_generic.hip dft8)
__device__ __forceinline__ void dft8(f2 (&v)[8])
{
    const float h = 0.70710678118654752f;
    f2 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        a[j] = v[j] + v[j + 4];
        b[j] = v[j] - v[j + 4];
    }
    b[1] = f2{h * (b[1].x + b[1].y), h * (b[1].y - b[1].x)};      // (1 - i)/sqrt2
    b[2] = f2{b[2].y, -b[2].x};                                   // -i
    b[3] = f2{h * (b[3].y - b[3].x), -h * (b[3].x + b[3].y)};     // (-1 - i)/sqrt2
    auto dft4 = [](const f2 (&u)[4], f2 &y0, f2 &y1, f2 &y2, f2 &y3) {
        const f2 p0 = u[0] + u[2], p1 = u[0] - u[2], q0 = u[1] + u[3];
        const f2 d = u[1] - u[3], q1 = f2{d.y, -d.x};             // (u1 - u3) . (-i)
        y0 = p0 + q0;
        y1 = p1 + q1;
        y2 = p0 - q0;
        y3 = p1 - q1;
    };
    dft4(a, v[0], v[2], v[4], v[6]);
    dft4(b, v[1], v[3], v[5], v[7]);
}

// The two transposes between the radix-8 passes, in registers.  A wave holds a frame as 8 complex registers x 64 lanes; pass 2
// wants register index <-> lane bits 3..5 exchanged, pass 3 register index <-> lane bits 0..2.  An 8 x 8 transpose is three
// rounds of 2 x 2 block swaps (register bit s against one lane bit): for a register pair (a, b) the lanes whose bit is 1 take
// the partner lane's b into a, the lanes whose bit is 0 take the partner's a into b.  gfx950 has the swap itself for lane bits
// 4 and 5 (v_permlane16_swap / v_permlane32_swap: one instruction a pair), masked DPP row shifts do bits 2 and 3 (two
// instructions a pair), bits 0 and 1 take a quad permute and a select.  ~110 vector instructions replace 16 KB of LDS traffic
// and two LDS round trips per frame.
template <int CTRL_A, int BANK_A, int CTRL_B, int BANK_B>
__device__ __forceinline__ void swap_dpp_masked(float &a, float &b)
{
    const int na = __builtin_amdgcn_update_dpp((int)__float_as_uint(a), (int)__float_as_uint(b), CTRL_A, 0xF, BANK_A, false);
    const int nb = __builtin_amdgcn_update_dpp((int)__float_as_uint(b), (int)__float_as_uint(a), CTRL_B, 0xF, BANK_B, false);
    a = __uint_as_float((unsigned)na);
    b = __uint_as_float((unsigned)nb);
}
template <int CTRL>
__device__ __forceinline__ void swap_dpp_select(float &a, float &b, bool bit)
{
    const float pb = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(b), CTRL, 0xF, 0xF, false));
    const float pa = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(a), CTRL, 0xF, 0xF, false));
    a = bit ? pb : a;
    b = bit ? b : pa;
}
__device__ __forceinline__ void transpose_hi(f2 (&v)[8])          // register bits 0, 1, 2 <-> lane bits 3, 4, 5
{
#pragma unroll
    for (int r = 0; r < 8; r += 2) {                              // lane bit 3: l ^ 8 = row_ror:8; lanes 8-15 of a row are banks 2, 3
        float ax = v[r].x, ay = v[r].y, bx = v[r + 1].x, by = v[r + 1].y;
        swap_dpp_masked<0x128, 0xC, 0x128, 0x3>(ax, bx);
        swap_dpp_masked<0x128, 0xC, 0x128, 0x3>(ay, by);
        v[r] = f2{ax, ay};
        v[r + 1] = f2{bx, by};
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {                                 // lane bit 4
        const int r = (k & 1) + 4 * (k >> 1);
        auto sx = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[r].x), __float_as_uint(v[r + 2].x), false, false);
        auto sy = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[r].y), __float_as_uint(v[r + 2].y), false, false);
        v[r] = f2{__uint_as_float(sx[0]), __uint_as_float(sy[0])};
        v[r + 2] = f2{__uint_as_float(sx[1]), __uint_as_float(sy[1])};
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {                                 // lane bit 5
        auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[r].x), __float_as_uint(v[r + 4].x), false, false);
        auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[r].y), __float_as_uint(v[r + 4].y), false, false);
        v[r] = f2{__uint_as_float(sx[0]), __uint_as_float(sy[0])};
        v[r + 4] = f2{__uint_as_float(sx[1]), __uint_as_float(sy[1])};
    }
}
__device__ __forceinline__ void transpose_lo(f2 (&v)[8], int lane)   // register bits 0, 1, 2 <-> lane bits 0, 1, 2
{
    const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll
    for (int r = 0; r < 8; r += 2) {                              // lane bit 0: quad_perm [1, 0, 3, 2]
        float ax = v[r].x, ay = v[r].y, bx = v[r + 1].x, by = v[r + 1].y;
        swap_dpp_select<0xB1>(ax, bx, b0);
        swap_dpp_select<0xB1>(ay, by, b0);
        v[r] = f2{ax, ay};
        v[r + 1] = f2{bx, by};
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {                                 // lane bit 1: quad_perm [2, 3, 0, 1]
        const int r = (k & 1) + 4 * (k >> 1);
        float ax = v[r].x, ay = v[r].y, bx = v[r + 2].x, by = v[r + 2].y;
        swap_dpp_select<0x4E>(ax, bx, b1);
        swap_dpp_select<0x4E>(ay, by, b1);
        v[r] = f2{ax, ay};
        v[r + 2] = f2{bx, by};
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {                                 // lane bit 2: lanes 4-7, 12-15 (banks 1, 3) read l - 4 (row_shr:4), the others l + 4
        float ax = v[r].x, ay = v[r].y, bx = v[r + 4].x, by = v[r + 4].y;
        swap_dpp_masked<0x114, 0xA, 0x104, 0x5>(ax, bx);
        swap_dpp_masked<0x114, 0xA, 0x104, 0x5>(ay, by);
        v[r] = f2{ax, ay};
        v[r + 4] = f2{bx, by};
    }
}

// Which of the two goes through registers (A/B builds: -DSYLDET_FFT1K_T=0 both through LDS, 1 the first, 2 the second, 3 both)
#ifndef SYLDET_FFT1K_T
#define SYLDET_FFT1K_T 1
#endif
constexpr bool kRegT1 = (SYLDET_FFT1K_T & 1) != 0, kRegT2 = (SYLDET_FFT1K_T & 2) != 0;
// Software pipeline (with the first transpose in registers; -DSYLDET_FFT1K_PIPE=1, an experiment): the next frame's first
// pass -- window, radix-8, twiddles, the register transpose: vector work only -- is issued while this frame's two LDS round
// trips are in flight (the counters have the waves of this kernel waiting 41 % of their cycles).  Measured slower: 3.18 ms
// against 2.98 for the frame-at-a-time loop on one box (168 registers against 147, the same three waves a SIMD).
#ifndef SYLDET_FFT1K_PIPE
#define SYLDET_FFT1K_PIPE 0
#endif
constexpr bool kPipe = SYLDET_FFT1K_PIPE != 0 && kRegT1 && !kRegT2 && kFly == 1;

// AL: every frame starts 8-byte aligned (sample pairs in one load); otherwise -- odd hop, gap or row length -- two loads a point.
template <int KB, bool AL>
__global__ void __launch_bounds__(kBlock)
fft1k_net_kernel(const StftDesc sd_, const MlpxDesc d, const float *__restrict__ samples, int64_t stride, int64_t J, int64_t E,
                 int tiles_per_channel, int tiles_per_run, float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: first-layer fragments | columns hi | columns lo | per-frame sums, exponents | transform scratch, whose first part
    // holds the tap products once the tile's transforms are done (a barrier apart)
    const int CS = d.col_stride, PS = d.p_stride;
    uint32x4 *afr = reinterpret_cast<uint32x4 *>(smem);
    _Float16 *colh = reinterpret_cast<_Float16 *>(smem + 3 * KB * 2 * 1024);
    _Float16 *coll = colh + kTile * CS;
    float *ssf = reinterpret_cast<float *>(coll + kTile * CS);       // [tile] per-frame sums of squares (true units)
    float *fsc = ssf + kTile;                                        // [tile] 2^-fe: a frame's products back to true units
    f2 *scratch = reinterpret_cast<f2 *>(fsc + kTile);               // [waves][kFly][kScratch]
    float *pbuf = reinterpret_cast<float *>(scratch);                // [tile][PS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = lane & 15, g4 = lane >> 4;
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const int c = blockIdx.y;
    const int F = d.F, T = d.T;
    const int step = kTile - (T - 1);                 // evaluations per tile = frames a tile advances by
    const float *chan = samples + (int64_t)c * stride + sd_.gap;
    f2 *buf0 = scratch + wave * kFly * kScratch;

    // ---- once per workgroup: the folded first layer's fragments -> LDS; zeros in the padding bins F .. 32 KB - 1 of every
    // column row (the weights there are zero too, but 0 * NaN from stale LDS is not)
    for (int i = tid; i < 3 * KB * 2 * 64; i += kBlock) afr[i] = reinterpret_cast<const uint32x4 *>(d.afrag)[i];
    for (int i = tid; i < kTile * (32 * KB - F); i += kBlock) {
        const int per = 32 * KB - F, fr = i / per, bin = F + (i - fr * per);
        colh[fr * CS + bin] = (_Float16)0.0f;
        coll[fr * CS + bin] = (_Float16)0.0f;
    }
    const float b0[4] = {d.bias0[0], d.bias0[1], d.bias0[2], d.bias0[3]}, w1[4] = {d.w1[0], d.w1[1], d.w1[2], d.w1[3]};
    const double thr = d.thresholds[0];
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(outputs ? outputs + (int64_t)c * E : nullptr, 0, outputs ? (int)(E * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t flg_rs = __builtin_amdgcn_make_buffer_rsrc(flags ? flags + (int64_t)c * E : nullptr, 0, flags ? (int)E : 0, 0x00020000);

    // ---- per-lane constants of the transform: window for points 64 a + lane (as even / odd pairs), twiddles of the two
    // inter-pass multiplications, split twiddles of this lane's two bins
    auto w1024 = [&](int idx) {                                   // e^{-2 pi i idx / 1024} from the half table
        idx &= 1023;
        const float2 w = sd_.sw[idx & 511];
        return idx & 512 ? f2{-w.x, -w.y} : f2{w.x, w.y};
    };
    f2 win[8], tw1[8], tw2[8];
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const float2 w = reinterpret_cast<const float2 *>(sd_.window)[64 * a + lane];
        win[a] = f2{w.x, w.y};
        tw1[a] = w1024(2 * lane * a);                             // W512^(b c), b = lane, c = a
        tw2[a] = w1024(16 * lo3 * a);                             // W64^(b' c'), b' = lane & 7, c' = a
    }
    auto nat = [](int k) { return (k & 63) + 2 * ((k >> 3) & 7) + 80 * (k >> 6); };   // where Z[k] sits in the skewed natural-order image
    f2 swr[2];
    int kbin[2];
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int fb = lane + 64 * it;
        kbin[it] = fb < F ? sd_.f0 + fb : 1;                      // (lanes without a bin compute bin 1 and drop it)
        const float2 w = sd_.sw[kbin[it]];
        swr[it] = f2{w.x, w.y};
    }

    // which eighths of the spectrum the band (and its mirror image, for the real split) lives in: only those rows of the
    // last pass go back to LDS
    unsigned need = 0;
    for (int k = sd_.f0; k < sd_.f0 + F; k++) need |= (1u << (k >> 6)) | (1u << ((512 - k) >> 6));
    need = (unsigned)__builtin_amdgcn_readfirstlane((int)need);

    const int tile0 = blockIdx.x * tiles_per_run;
    for (int tr = 0; tr < tiles_per_run; tr++) {
        const int tile = tile0 + tr;
        if (tile >= tiles_per_channel) break;                     // workgroup-uniform
        const int64_t e0 = (int64_t)tile * step;                  // first evaluation = first frame of the tile
        // ---- phase 1: the tile's new frames (all 128 in a run's first tile, the last 128 - (T-1) afterwards: the first T-1
        // rows were carried over), split over the waves in contiguous shares
        const int first_new = tr == 0 ? 0 : T - 1;
        const int n_new = kTile - first_new, per = (n_new + kWaves - 1) / kWaves;
        const int r0 = first_new + wave * per, r1 = (r0 + per < kTile) ? r0 + per : kTile;
        auto fetch = [&](int row, f2 (&raw)[8]) {                 // frames past the channel's end are clamped to its last one
            int64_t j = e0 + row;
            j = j < J ? j : J - 1;
            const float *xf = chan + j * sd_.hop;
            const float2 *x = reinterpret_cast<const float2 *>(xf);
#pragma unroll
            for (int a = 0; a < 8; a++) {
                if (AL) {
                    const float2 s = x[64 * a + lane];
                    raw[a] = f2{s.x, s.y};
                } else {
                    raw[a] = f2{xf[2 * (64 * a + lane)], xf[2 * (64 * a + lane) + 1]};
                }
            }
        };
        if (kPipe) {
            // p: the frame whose first pass is done; raw: the one behind it, on its way from memory
            f2 p[8], raw[8];
            auto first_pass = [&](f2 (&q)[8]) {                   // window (vDSP_vmul :311), even / odd packing (:314-316), over a -> c
#pragma unroll
                for (int a = 0; a < 8; a++) q[a] = raw[a] * win[a];
            };
            auto row_of = [&](int r) { return r < r1 ? r : (r1 > r0 ? r1 - 1 : r0); };
            fetch(row_of(r0), raw);
            first_pass(p);
            fetch(row_of(r0 + 1), raw);
            dft8(p);
#pragma unroll
            for (int cc = 1; cc < 8; cc++) p[cc] = cmul(p[cc], tw1[cc]);
            transpose_hi(p);                                      // lane = (c, b'), registers a': y[c][8 a' + b']
            for (int row = r0; row < r1; row++) {
                f2 *buf = buf0;
                f2 v[8], q[8];
#pragma unroll
                for (int a = 0; a < 8; a++) v[a] = p[a];
                dft8(v);                                          // over a' -> index c'
#pragma unroll
                for (int cc = 0; cc < 8; cc++) buf[(hi3 * 8 + cc) * 9 + lo3] = cc ? cmul(v[cc], tw2[cc]) : v[cc];
                // (in the shadow of that round trip: the next frame's window and radix-8)
                first_pass(q);
                fetch(row_of(row + 2), raw);
                dft8(q);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int a = 0; a < 8; a++) v[a] = buf[(hi3 * 8 + lo3) * 9 + a];   // lane = (c, c'): z[c][c'][b']
                __builtin_amdgcn_wave_barrier();
                dft8(v);                                          // over b' -> index d'
#pragma unroll
                for (int dd = 0; dd < 8; dd++)
                    if (need & (1u << dd)) buf[hi3 + 10 * lo3 + 80 * dd] = v[dd];
                // (and of this one: its twiddles and the register transpose)
#pragma unroll
                for (int cc = 1; cc < 8; cc++) q[cc] = cmul(q[cc], tw1[cc]);
                transpose_hi(q);
#pragma unroll
                for (int a = 0; a < 8; a++) p[a] = q[a];
                __builtin_amdgcn_wave_barrier();
                float cv[2], ss;
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    const f2 zk = buf[nat(kbin[it])], zm = buf[nat(512 - kbin[it])];
                    const float ar = zk.x + zm.x, ai = zk.y - zm.y, br = zk.x - zm.x, bi = zk.y + zm.y;
                    const float tre = br * swr[it].x - bi * swr[it].y, tim = br * swr[it].y + bi * swr[it].x;
                    const float re2 = ar + tim, im2 = ai - tre;
                    const float m = __builtin_amdgcn_sqrtf(re2 * re2 + im2 * im2) * 0.5f;      // zvabs / 2, :329-333
                    cv[it] = (lane + 64 * it < F) ? m : 0.0f;
                }
                ss = fmaf(cv[0], cv[0], cv[1] * cv[1]);
                __builtin_amdgcn_wave_barrier();
                {
                    float t = xor32_sum(xor16_sum(ss));
                    t += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(t), 0xB1, 0xF, 0xF, false));
                    t += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(t), 0x4E, 0xF, 0xF, false));
                    t += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(t), 0x141, 0xF, 0xF, false));
                    t += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(t), 0x140, 0xF, 0xF, false));
                    ss = t;
                }
                {
                    unsigned tb = ((__float_as_uint(ss) + 0x800000u) >> 1) & 0x7f800000u;
                    tb = (unsigned)min(max((int)tb, 16 << 23), 80 << 23);
                    const float up = __uint_as_float((203u << 23) - tb), down = __uint_as_float(tb + (51u << 23));
#pragma unroll
                    for (int it = 0; it < 2; it++) {
                        const int fb = lane + 64 * it;
                        if (fb < F) {
                            const float sc = cv[it] * up;
                            const _Float16 hh = (_Float16)sc;
                            colh[row * CS + fb] = hh;
                            coll[row * CS + fb] = (_Float16)(sc - (float)hh);
                        }
                    }
                    if (lane == 0) {
                        ssf[row] = ss;
                        fsc[row] = down;
                    }
                }
            }
        } else {
        // kFly frames at a time, stage by stage (rows past the share repeat its last row and are not stored)
        f2 nxt[kFly][8];
#pragma unroll
        for (int u = 0; u < kFly; u++) fetch(r0 + u < r1 ? r0 + u : (r1 > r0 ? r1 - 1 : r0), nxt[u]);
        for (int row = r0; row < r1; row += kFly) {
            f2 v[kFly][8];
#pragma unroll
            for (int u = 0; u < kFly; u++)
#pragma unroll
                for (int a = 0; a < 8; a++) v[u][a] = nxt[u][a] * win[a];   // window multiply (vDSP_vmul :311), even / odd packing (:314-316)
#pragma unroll
            for (int u = 0; u < kFly; u++) {                      // the next pair is on its way while this one is transformed
                const int nr = row + kFly + u;
                fetch(nr < r1 ? nr : r1 - 1, nxt[u]);
            }
#pragma unroll
            for (int u = 0; u < kFly; u++) {
                f2 *buf = buf0 + u * kScratch;
                dft8(v[u]);                                       // over a -> index c
                if (kRegT1) {
#pragma unroll
                    for (int cc = 1; cc < 8; cc++) v[u][cc] = cmul(v[u][cc], tw1[cc]);
                    transpose_hi(v[u]);                           // lane = (c, b'), registers a': y[c][8 a' + b']
                } else {
#pragma unroll
                    for (int cc = 0; cc < 8; cc++) buf[cc * 72 + lane] = cc ? cmul(v[u][cc], tw1[cc]) : v[u][cc];
                }
            }
            if (!kRegT1) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < kFly; u++) {
                    const f2 *buf = buf0 + u * kScratch;
#pragma unroll
                    for (int a = 0; a < 8; a++) v[u][a] = buf[hi3 * 72 + 8 * a + lo3];    // lane = (c, b'): y[c][8 a' + b']
                }
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int u = 0; u < kFly; u++) {
                f2 *buf = buf0 + u * kScratch;
                dft8(v[u]);                                       // over a' -> index c'
                if (kRegT2) {
#pragma unroll
                    for (int cc = 1; cc < 8; cc++) v[u][cc] = cmul(v[u][cc], tw2[cc]);
                    transpose_lo(v[u], lane);                     // lane = (c, c'), registers b': z[c][c'][b']
                } else {
#pragma unroll
                    for (int cc = 0; cc < 8; cc++) buf[(hi3 * 8 + cc) * 9 + lo3] = cc ? cmul(v[u][cc], tw2[cc]) : v[u][cc];
                }
            }
            if (!kRegT2) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < kFly; u++) {
                    const f2 *buf = buf0 + u * kScratch;
#pragma unroll
                    for (int a = 0; a < 8; a++) v[u][a] = buf[(hi3 * 8 + lo3) * 9 + a];   // lane = (c, c'): z[c][c'][b']
                }
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int u = 0; u < kFly; u++) {
                f2 *buf = buf0 + u * kScratch;
                dft8(v[u]);                                       // over b' -> index d'
#pragma unroll
                // Z[c + 8 c' + 64 d'] in natural order, every row of eight skewed by two elements (a lane group of 16 holds two
                // values of c and all eight of c': without the skew the eight land on two sets of banks, a 4-way conflict)
                for (int dd = 0; dd < 8; dd++)
                    if (need & (1u << dd)) buf[hi3 + 10 * lo3 + 80 * dd] = v[u][dd];
            }
            __builtin_amdgcn_wave_barrier();
            // real split + |X| of this lane's bins (2X[k] = (Z[k] + conj Z[M-k]) - i e^{-2 pi i k/N} (Z[k] - conj Z[M-k]),
            // :320-333; bins stay above 0 and below N/2 in this kernel's class)
            float cv[kFly][2], ss[kFly];
#pragma unroll
            for (int u = 0; u < kFly; u++) {
                const f2 *buf = buf0 + u * kScratch;
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    const f2 zk = buf[nat(kbin[it])], zm = buf[nat(512 - kbin[it])];
                    const float ar = zk.x + zm.x, ai = zk.y - zm.y, br = zk.x - zm.x, bi = zk.y + zm.y;
                    const float tre = br * swr[it].x - bi * swr[it].y, tim = br * swr[it].y + bi * swr[it].x;
                    const float re2 = ar + tim, im2 = ai - tre;
                    const float m = __builtin_amdgcn_sqrtf(re2 * re2 + im2 * im2) * 0.5f;      // zvabs / 2, :329-333
                    cv[u][it] = (lane + 64 * it < F) ? m : 0.0f;
                }
                ss[u] = fmaf(cv[u][0], cv[u][0], cv[u][1] * cv[u][1]);
            }
            __builtin_amdgcn_wave_barrier();
            // the frame's sum of squares and its own column exponent: the column goes to LDS as f16 hi + lo at the scale that
            // puts its norm into [2^12, 2^13) (see kernels_fused_r.hip, mag_micro)
#pragma unroll
            for (int u = 0; u < kFly; u++) {
                float t = xor32_sum(xor16_sum(ss[u]));
                t += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(t), 0xB1, 0xF, 0xF, false));
                t += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(t), 0x4E, 0xF, 0xF, false));
                t += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(t), 0x141, 0xF, 0xF, false));
                t += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(t), 0x140, 0xF, 0xF, false));
                ss[u] = t;
            }
#pragma unroll
            for (int u = 0; u < kFly; u++) {
                const int rw = row + u;
                if (rw < r1) {                                    // wave-uniform
                    unsigned tb = ((__float_as_uint(ss[u]) + 0x800000u) >> 1) & 0x7f800000u;
                    tb = (unsigned)min(max((int)tb, 16 << 23), 80 << 23);
                    const float up = __uint_as_float((203u << 23) - tb), down = __uint_as_float(tb + (51u << 23));
#pragma unroll
                    for (int it = 0; it < 2; it++) {
                        const int fb = lane + 64 * it;
                        if (fb < F) {
                            const float sc = cv[u][it] * up;
                            const _Float16 hh = (_Float16)sc;
                            colh[rw * CS + fb] = hh;
                            coll[rw * CS + fb] = (_Float16)(sc - (float)hh);
                        }
                    }
                    if (lane == 0) {
                        ssf[rw] = ss[u];
                        fsc[rw] = down;
                    }
                }
            }
        }
        }
        __syncthreads();
        // ---- phase 2: tap products of this wave's 16 frames, P[(t, h), j] for all taps at once (three row tiles), back to
        // true units by the frame's exponent
        if (16 * wave < kTile) {
            const int fr = 16 * wave + f;
            const _Float16 *bph = colh + fr * CS + 8 * g4, *bpl = coll + fr * CS + 8 * g4;
            floatx4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kb = 0; kb < KB; kb++) {
                const half8 bh = as_half8(*reinterpret_cast<const uint32x4 *>(bph + 32 * kb));
                const half8 bl = as_half8(*reinterpret_cast<const uint32x4 *>(bpl + 32 * kb));
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    const half8 ah = as_half8(afr[((m * KB + kb) * 2 + 0) * 64 + lane]), al = as_half8(afr[((m * KB + kb) * 2 + 1) * 64 + lane]);
                    acc[m] = mfma(ah, bh, acc[m]);
                    acc[m] = mfma(ah, bl, acc[m]);
                    acc[m] = mfma(al, bh, acc[m]);
                }
            }
            const float dn = fsc[fr];
#pragma unroll
            for (int m = 0; m < 3; m++) *reinterpret_cast<floatx4 *>(pbuf + fr * PS + 4 * (4 * m + g4)) = acc[m] * dn;
        }
        __syncthreads();
        // ---- phase 3: evaluations e0 .. e0 + step - 1, one thread each: diagonal sum over the taps, l2normalize
        // (NeuralNet.swift:47-59), the rest of the network
        if (tid < step) {
            floatx4 z = {0.f, 0.f, 0.f, 0.f};
            float ssw = 0.0f;
            for (int t = 0; t < T; t++) {
                z += *reinterpret_cast<const floatx4 *>(pbuf + (tid + t) * PS + 4 * t);
                ssw += ssf[tid + t];
            }
            const float alpha = d.w_unscale * __builtin_amdgcn_rsqf(ssw);
            float y = d.b1;
#pragma unroll
            for (int j = 0; j < 4; j++) {             // TanSig hidden units (rows past H meet zero weights), linear output
                const float a = fmaf(alpha, z[j], b0[j]);
                const float th = fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a * 2.885390081777927f) + 1.0f), 1.0f);
                y = fmaf(w1[j], th, y);
            }
            y = (y - d.oa) / d.og + d.ob;
            const int64_t e = e0 + tid;
            const bool st = e < E;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), out_rs, st ? (unsigned)e * 4u : 0xFFFFFFFFu, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)((double)y >= thr ? 1 : 0), flg_rs, st ? (unsigned)e : 0xFFFFFFFFu, 0, 0);
        }
        __syncthreads();                              // (the products are read: the scratch they sit in is free for the next tile's transforms)
        // ---- the last T - 1 frames are the next tile's first: their columns, sums and exponents move to the front
        if (tr + 1 < tiles_per_run) {
            const int words = (T - 1) * (CS / 2);     // 32-bit words per array
            for (int i = tid; i < 2 * words; i += kBlock) {
                unsigned *arr = reinterpret_cast<unsigned *>(i < words ? colh : coll);
                const int w = i < words ? i : i - words;
                arr[w] = arr[step * (CS / 2) + w];
            }
            if (tid < T - 1) {
                ssf[tid] = ssf[step + tid];
                fsc[tid] = fsc[step + tid];
            }
            __syncthreads();
        }
    }
}

}  // namespace

bool fft1k_applicable(const StftDesc &s, const MlpxDesc &d, const float *samples, int64_t stride)
{
    // 1024-point frames without zero padding, the band inside (0, N/2) and at most 128 bins (two per lane)
    (void)samples; (void)stride;
    return s.N == 1024 && s.W == 1024 && s.f0 >= 1 && s.f0 + s.F <= 511 && s.F <= 128 && s.power_mode == 0 &&
           d.KB == 4 && d.F == s.F && d.F % 4 == 0 && d.scaling == 0;
}

hipError_t launch_fft1k_net(const StftDesc &s, const MlpxDesc &d, const float *samples, int64_t stride, int C, int64_t J, int64_t E,
                            float *outputs, uint8_t *flags, hipStream_t stream)
{
    if (E <= 0 || C <= 0) return hipSuccess;
    if ((uint64_t)E * 4u >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    const int step = kTile - (d.T - 1);
    const int64_t tiles = (E + step - 1) / step;
    // a workgroup walks a contiguous run of tiles of one channel (the carried columns save T - 1 transforms a tile); runs as
    // long as still leaves a few workgroups per CU
    int64_t runs_per_channel = (1024 + C - 1) / C;
    runs_per_channel = runs_per_channel < 1 ? 1 : (runs_per_channel > tiles ? tiles : runs_per_channel);
    const int64_t tiles_per_run = (tiles + runs_per_channel - 1) / runs_per_channel;
    const int64_t runs = (tiles + tiles_per_run - 1) / tiles_per_run;
    dim3 grid((unsigned)runs, (unsigned)C);
    const bool aligned = (s.hop & 1) == 0 && (s.gap & 1) == 0 && (stride & 1) == 0 && (reinterpret_cast<uintptr_t>(samples) & 7) == 0;
    auto kern = aligned ? fft1k_net_kernel<4, true> : fft1k_net_kernel<4, false>;
    // LDS: the matrix-core stage's layout (fragments, columns, products, per-frame sums) with the transform's scratch where the
    // per-quad sums of the two-launch form were
    const int scratch = (kBlock / 64) * kFly * kScratch * 8, products = kTile * d.p_stride * 4;
    const int lds = 3 * 4 * 2 * 1024 + 2 * kTile * d.col_stride * 2 + 2 * kTile * 4 + (scratch > products ? scratch : products);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t st = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (st != hipSuccess) return st;
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), (size_t)lds, stream, s, d, samples, stride, J, E, (int)tiles, (int)tiles_per_run, outputs, flags);
    return hipGetLastError();
}

}  // namespace sd
