// kernels_stft_lanes.hip -- the generic engine's STFT for 128-, 256- and 512-point frames without an LDS pass between the
// butterflies: a frame is spread over P = 8, 16 or 32 lanes with eight points in each, the first radix-8 pass runs in
// registers and the remaining P-point transforms across the lanes (DPP row operations, v_permlane16_swap), one LDS round
// trip per frame for the real split.  Same path as stft_generic_kernel (kernels_generic.hip):
//   extractPower     Common/CircularShortTimeFourierTransform.swift:280-337   (window :311, packing :314-316, real FFT
//                                                                              :317-320, Nyquist dropped :323, zvabs / 2 :329-333)
//   extractMagnitude :221-278                                                  (zvmags / 4 :270-274)
//
//   packed frame z[m] = xw[2m] + i xw[2m+1], m = P a + b < M = 8 P   (a: register, b: lane of the frame)
//   Z[c + 8 d] = sum_b W_P^(b d) . W_M^(b c) . sum_a z[P a + b] W_8^(a c)
//     1. eight-point transforms over a, in registers                      -> y[c][b]
//     2. twiddles W_M^(b c)
//     3. P-point transforms over b, across the lanes: log2 P radix-2 decimation-in-frequency stages -- lane b trades with
//        lane b ^ S, u = (value of the lane with bit S clear) +- (value of the lane with it set), the latter lane multiplies
//        by W_2S^(b mod S) -- which leave index d = bitrev(b) in lane b
//     4. Z to LDS in natural order (rows of eight padded to nine: a lane group's stores hit distinct banks), real split
//        2X[k] = (Z[k] + conj Z[M-k]) - i e^{-2 pi i k/N} (Z[k] - conj Z[M-k]) and |X| for the band's bins, P per lane round.
//
// gfx950 only.  wave = 64.

#include <cstdlib>

#include "kernels.hpp"

namespace sd {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr int kRounds = 32;                // frame groups per wave: the per-lane tables are set up once for 32 * 64 / P frames

__device__ __forceinline__ f2 cmul(f2 a, f2 b) { return f2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

__device__ __forceinline__ void dft8(f2 (&v)[8])
{
    // radix-2 decimation in frequency: X[2m] from a = v[j] + v[j+4], X[2m+1] from b = (v[j] - v[j+4]) W8^j, each a 4-point DFT
    const float h = 0.70710678118654752f;
    f2 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        a[j] = v[j] + v[j + 4];
        b[j] = v[j] - v[j + 4];
    }
    b[1] = f2{h * (b[1].x + b[1].y), h * (b[1].y - b[1].x)};        // (1 - i)/sqrt2
    b[2] = f2{b[2].y, -b[2].x};                                     // -i
    b[3] = f2{h * (b[3].y - b[3].x), -h * (b[3].x + b[3].y)};       // (-1 - i)/sqrt2
    auto dft4 = [](const f2 (&u)[4], f2 &y0, f2 &y1, f2 &y2, f2 &y3) {
        const f2 p0 = u[0] + u[2], p1 = u[0] - u[2], q0 = u[1] + u[3];
        const f2 q1 = f2{u[1].y - u[3].y, -(u[1].x - u[3].x)};      // (u1 - u3) . (-i)
        y0 = p0 + q0;
        y1 = p1 + q1;
        y2 = p0 - q0;
        y3 = p1 - q1;
    };
    dft4(a, v[0], v[2], v[4], v[6]);
    dft4(b, v[1], v[3], v[5], v[7]);
}

// the value of lane ^ S (within the frame's lanes), no register tied to the result
template <int S>
__device__ __forceinline__ float partner(float x)
{
    const int xi = (int)__float_as_uint(x);
    if (S == 1) return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp(xi, 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    if (S == 2) return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp(xi, 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    if (S == 4) {                                                                  // lanes 4k..4k+3 of a row are a bank
        int p = __builtin_amdgcn_mov_dpp(xi, 0x104, 0xF, 0x5, false);             // row_shl:4 -> banks 0, 2 read lane + 4
        p = __builtin_amdgcn_update_dpp(p, xi, 0x114, 0xF, 0xA, false);           // row_shr:4 -> banks 1, 3 read lane - 4
        return __uint_as_float((unsigned)p);
    }
    return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp(xi, 0x128, 0xF, 0xF, true));                  // S == 8: row_ror:8
}

// One decimation-in-frequency stage across the lanes, for the eight values of a lane at once (so that no result is read by
// the instruction behind the one that wrote it): u = (the value of the lane with bit S clear) + sgn (that of the lane with
// it set), sgn = +1 where this lane's bit is clear and -1 where it is set; then the twiddle (1 in the lanes with the bit clear).
template <int S, bool TW>
__device__ __forceinline__ void stage(f2 (&v)[8], float sgn, f2 tw)
{
    if (S == 32) {                                              // (likewise with the wave's halves)
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[c].x), __float_as_uint(v[c].x), false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[c].y), __float_as_uint(v[c].y), false, false);
            v[c] = f2{fmaf(__uint_as_float(rx[1]), sgn, __uint_as_float(rx[0])), fmaf(__uint_as_float(ry[1]), sgn, __uint_as_float(ry[0]))};
        }
    } else if (S == 16) {
        // v_permlane16_swap(a, b): a's odd rows <-> b's even rows.  With a = b = x: r[0] holds the even rows' values in both
        // rows of a pair, r[1] the odd rows' -- the (lo, hi) of this stage in every lane
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const auto rx = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[c].x), __float_as_uint(v[c].x), false, false);
            const auto ry = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[c].y), __float_as_uint(v[c].y), false, false);
            v[c] = f2{fmaf(__uint_as_float(rx[1]), sgn, __uint_as_float(rx[0])), fmaf(__uint_as_float(ry[1]), sgn, __uint_as_float(ry[0]))};
        }
    } else {
        // (four values at a time: enough independent work between a result and the instruction that reads it across lanes,
        // without sixteen more live registers)
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
            f2 t[4];
#pragma unroll
            for (int c = 0; c < 4; c++) t[c] = v[h + c] * sgn;
#pragma unroll
            for (int c = 0; c < 4; c++) v[h + c] = f2{partner<S>(v[h + c].x) + t[c].x, partner<S>(v[h + c].y) + t[c].y};
        }
    }
    if (TW) {
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = cmul(v[c], tw);
    }
}

// L = log2 P.  Frames per wave and round: 64 / P.  PAD = false: every frame starts 8-byte aligned and the window fills the
// frame (the launcher checks) -- a lane's points come as (even, odd) sample pairs in one load; PAD = true: any alignment and
// any window up to N (zero pad :110), two loads a point, clamped into the window (what they fetch past it meets a zero).
template <int L, bool PAD>
__global__ void __launch_bounds__(kBlock, L >= 5 ? 3 : 4)      // (512-point frames at four waves a SIMD spill 16 registers: 4.87 ms against 4.17)
stft_lanes_kernel(StftDesc d, const float *__restrict__ samples, int64_t stride, int64_t J, float *__restrict__ columns)
{
    constexpr int P = 1 << L, FW = kWave / P, M = 8 * P;
    constexpr int kScratch = FW * 9 * P;                        // f2 per wave: rows of eight padded to nine
    __shared__ f2 lds_all[(kBlock / kWave) * kScratch];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int b = lane & (P - 1), fw = lane >> L;
    f2 *buf = lds_all + wave * kScratch + fw * 9 * P;
    const int c = blockIdx.y;
    const float *chan = samples + (int64_t)c * stride + d.gap;
    float *cols = columns + (int64_t)c * J * d.F;
    const int N = 2 * M;
    auto wN = [&](int idx) {                                    // e^{-2 pi i idx / N} from the half table sw[k], k < M
        idx &= N - 1;
        const float2 w = d.sw[idx & (M - 1)];
        return idx & M ? f2{-w.x, -w.y} : f2{w.x, w.y};
    };
    // per-lane constants: window of points P a + b (even / odd sample of the pair; zero past the window: the zero pad :110 --
    // the loads there are clamped into the window, whatever they fetch meets a zero), twiddles between the passes, the
    // stages' signs and twiddles, where this lane's results go
    f2 win[8], tw1[8];
    int moff[8];
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const int m = P * a + b, n0 = 2 * m;
        win[a] = f2{n0 < d.W ? d.window[n0] : 0.0f, n0 + 1 < d.W ? d.window[n0 + 1] : 0.0f};
        moff[a] = PAD ? (n0 < d.W ? n0 : d.W - 1) : m;          // (PAD: the even sample's index, clamped into the window)
        tw1[a] = wN(2 * b * a);                                 // W_M^(b c), c = a
    }
    float sgn[6];
    f2 tws[6];
#pragma unroll
    for (int s = 0; s < 6; s++) {
        const int S = 1 << s;
        const bool up = (b & S) != 0;
        sgn[s] = up ? -1.0f : 1.0f;
        tws[s] = (s < L && up) ? wN((b & (S - 1)) * (N / (2 * S))) : f2{1.0f, 0.0f};     // W_2S^(b mod S)
    }
    int brev = 0;
#pragma unroll
    for (int s = 0; s < L; s++) brev |= ((b >> s) & 1) << (L - 1 - s);
    const int pos0 = 9 * brev;                                  // Z[c + 8 brev] sits at 9 brev + c
    auto pos = [](int k) { return 9 * (k >> 3) + (k & 7); };
    // this lane's bins f = b + P it: where Z[k] and Z[M - k] sit and the split twiddle, for the first two rounds of bins in
    // registers (bands of up to 2 P bins), fetched when a wider band needs more
    constexpr int kBinRegs = 2;
    int pk[kBinRegs], pm[kBinRegs];
    f2 swr[kBinRegs];
#pragma unroll
    for (int it = 0; it < kBinRegs; it++) {
        const int f = b + P * it, k = d.f0 + (f < d.F ? f : 0);
        pk[it] = pos(k);
        pm[it] = pos(k ? M - k : 0);
        const float2 w = d.sw[k];
        swr[it] = f2{w.x, w.y};
    }

    const int64_t j0 = ((int64_t)blockIdx.x * (kBlock / kWave) + wave) * (kRounds * FW);
    if (j0 >= J) return;
    auto fetch = [&](int64_t j, f2 (&raw)[8]) {
        j = j < J ? j : J - 1;                                  // frames past the end repeat the last one and are not stored
        const float *xf = chan + j * d.hop;
        const float2 *x = reinterpret_cast<const float2 *>(xf);
#pragma unroll
        for (int a = 0; a < 8; a++) {
            if (PAD) {                                          // any alignment, any window up to N: two loads a point
                const int i1 = moff[a] + 1 < d.W ? moff[a] + 1 : d.W - 1;
                raw[a] = f2{xf[moff[a]], xf[i1]};
            } else {
                const float2 s2 = x[P * a + b];
                raw[a] = f2{s2.x, s2.y};
            }
        }
    };
    f2 nxt[8];
    fetch(j0 + fw, nxt);
    for (int r = 0; r < kRounds; r++) {
        const int64_t j = j0 + (int64_t)r * FW + fw;
        if (j0 + (int64_t)r * FW >= J) return;                  // wave-uniform
        f2 v[8];
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = nxt[a] * win[a];     // window multiply (vDSP_vmul :311), even / odd packing (:314-316)
        if (r + 1 < kRounds) fetch(j + FW, nxt);
        dft8(v);                                                // over a -> index c
#pragma unroll
        for (int cc = 1; cc < 8; cc++) v[cc] = cmul(v[cc], tw1[cc]);
        // the P-point transforms over b, largest span first
        if (L > 5) stage<32, true>(v, sgn[5], tws[5]);
        if (L > 4) stage<16, true>(v, sgn[4], tws[4]);
        if (L > 3) stage<8, true>(v, sgn[3], tws[3]);
        stage<4, true>(v, sgn[2], tws[2]);
        stage<2, true>(v, sgn[1], tws[1]);                      // (1 or -i)
        stage<1, false>(v, sgn[0], tws[0]);
#pragma unroll
        for (int cc = 0; cc < 8; cc++) buf[pos0 + cc] = v[cc];  // Z[c + 8 bitrev(b)], natural order
        __builtin_amdgcn_wave_barrier();
        // real split + magnitude for the band only; Nyquist is dropped (:323)
        auto split_bin = [&](int f, f2 zk, f2 zm, f2 w) {
            const int k = d.f0 + f;
            const float ar = zk.x + zm.x, ai = zk.y - zm.y;
            const float br = zk.x - zm.x, bi = zk.y + zm.y;
            const float tr = br * w.x - bi * w.y, ti = br * w.y + bi * w.x;
            const float re2 = k ? ar + ti : 2.0f * (zk.x + zk.y), im2 = k ? ai - tr : 0.0f;      // (bin 0: X[0] = Re Z[0] + Im Z[0])
            const float p = re2 * re2 + im2 * im2;
            // zvmags/4 :270-274, zvabs/2 :329-333 (the hardware square root: 1 ulp, and NaN / inf / 0 as the library's)
            if (j < J && f < d.F) cols[j * d.F + f] = d.power_mode ? p * 0.25f : __builtin_amdgcn_sqrtf(p) * 0.5f;
        };
#pragma unroll
        for (int it = 0; it < kBinRegs; it++)
            if (P * it < d.F) split_bin(b + P * it, buf[pk[it]], buf[pm[it]], swr[it]);           // (wave-uniform test)
        for (int f = b + P * kBinRegs; f < d.F; f += P) {
            const int k = d.f0 + f;
            const float2 w = d.sw[k];
            split_bin(f, buf[pos(k)], buf[pos(M - k)], f2{w.x, w.y});
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

bool stft_lanes_applicable(const StftDesc &d, const float *samples, int64_t stride)
{
    // 128-, 256- or 512-point frames
    (void)samples; (void)stride;
    static const bool with_1k = std::getenv("SYLDET_LANES_1K") != nullptr;      // (experiment: 1024-point frames too, instead of stft_r8_kernel)
    return (d.M == 64 || d.M == 128 || d.M == 256 || (with_1k && d.M == 512)) && d.W <= 2 * d.M && d.W >= 1 && d.F >= 1 && d.f0 >= 0 && d.f0 + d.F <= d.M;
}

hipError_t launch_stft_lanes(const StftDesc &d, const float *samples, int64_t stride, int C, int64_t J, float *columns, hipStream_t stream)
{
    if (J <= 0 || C <= 0) return hipSuccess;
    if (!stft_lanes_applicable(d, samples, stride)) return hipErrorInvalidValue;
    const int P = d.M / 8;
    const int64_t per_block = (int64_t)(kBlock / kWave) * kRounds * (kWave / P);
    dim3 grid((unsigned)((J + per_block - 1) / per_block), (unsigned)C);
    // sample pairs in one load where every frame is 8-byte aligned and the window fills it
    const bool pad = d.W < 2 * d.M || (d.hop & 1) != 0 || (d.gap & 1) != 0 || (stride & 1) != 0 || (reinterpret_cast<uintptr_t>(samples) & 7) != 0;
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(kBlock), 0, stream, d, samples, stride, J, columns); };
    if (d.M == 64) pad ? go(stft_lanes_kernel<3, true>) : go(stft_lanes_kernel<3, false>);
    else if (d.M == 128) pad ? go(stft_lanes_kernel<4, true>) : go(stft_lanes_kernel<4, false>);
    else if (d.M == 256) pad ? go(stft_lanes_kernel<5, true>) : go(stft_lanes_kernel<5, false>);
    else pad ? go(stft_lanes_kernel<6, true>) : go(stft_lanes_kernel<6, false>);
    return hipGetLastError();
}

}  // namespace sd
