// kernels_mlpx.hip -- the network stage of the generic engine on the matrix cores.
//
// Reference path being replaced (reference root relative): SyllableDetector.processNewValue's window of timeRange
// columns (Common/SyllableDetector.swift:153-217) -> NeuralNet.apply (Common/NeuralNet.swift:294-326, :366-377;
// L2Normalize :47-59, MapMinMax / MapStd :127-131 / :162-169 and their reverse maps, TanSig :189-194) -> lastDetected
// (Common/SyllableDetector.swift:27-31), for spectrogram columns that are already in HBM ([C][J][F] fp32, written by the
// generic STFT kernels: long windows, wide bands -- BASELINE configs[2]).
//
// Formulation: with the affine input maps folded into the first layer on the host, the layer's input to unit h for
// evaluation e is  sum_t W'_t[h, :] . c(e + t)  over the timeRange columns c of its window.  Every column meets every
// tap exactly once, so ALL taps are the rows of one plain GEMM
//     P[(t, h), j] = W'_t[h, :] . c(j)          (rows 4 t + h: 12 taps x 4 units = three 16-row tiles; K = bins; N = frames)
// on the matrix cores (v_mfma_f32_16x16x32_f16; every operand f16 hi + lo, hi*hi + hi*lo + lo*hi reproduces the fp32
// product, fp32 accumulate), and an evaluation is the diagonal sum  sum_t P[(t, h), e + t]  read back from LDS.  Against
// one GEMM per tap (the fused kernels' shape: 4 of 16 rows used, every column fetched timeRange times) that is 3.3x
// fewer MFMAs and 20x less LDS traffic.
//
// A tile = 128 frames of one channel (16 per wave): read once as coalesced quads (the next tile's are in flight meanwhile),
// every frame scaled by its own power of two (its column norm goes to [2^12, 2^13): a quiet frame next to a loud one keeps 22
// bits of its own level), split into f16 hi + lo in LDS; the l2normalize denominator comes from sums of squares taken per
// quad and added per frame in a fixed order (no atomics: results do not depend on the order the threads arrive in); tap
// products go back to true units by the frame's exponent and to LDS;
// the tile's 128 - timeRange + 1 evaluations finish in registers, one thread each.
//
// gfx950 only.  wave = 64.

#include "fused_common.hpp"

namespace sd {

namespace {

using namespace fused_dev;

constexpr int kBlock = kMlpxBlock;
constexpr int kTile = kMlpxTile;

template <int KB>
__global__ void __launch_bounds__(kBlock, 2)
mlp_mfma_kernel(const MlpxDesc d, const float *__restrict__ columns, int64_t J, int64_t E, int tiles_per_channel,
                float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32x4 *afr = reinterpret_cast<uint32x4 *>(smem + d.lds_afrag);
    _Float16 *colh = reinterpret_cast<_Float16 *>(smem + d.lds_colh);
    _Float16 *coll = reinterpret_cast<_Float16 *>(smem + d.lds_coll);
    float *pbuf = reinterpret_cast<float *>(smem + d.lds_p);
    float *pq = reinterpret_cast<float *>(smem + d.lds_pq);          // [frame][F / 4] sums of squares per quad
    float *ss0 = reinterpret_cast<float *>(smem + d.lds_ss);         // [2 tiles][128] sums of squares per frame
    float *fup = reinterpret_cast<float *>(smem + d.lds_red);        // [128] per-frame scale 2^fe and, behind it, its inverse
    float *fdn = fup + kTile;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = lane & 15, g4 = lane >> 4;
    const int c = blockIdx.y;
    const int F = d.F, T = d.T, CS = d.col_stride, PS = d.p_stride;
    const int step = kTile - (T - 1);                 // evaluations per tile = frames a tile advances by
    // a frame's bins as Fq quads; when F is not a multiple of 4 the last one reaches into the next frame's first bins (the
    // columns buffer ends 16 bytes after its last value for that) and is masked after the load
    const int Fq = (F + 3) / 4;
    const int nq = kTile * Fq;                        // quads of column values per tile
    const float *chan = columns + (int64_t)c * J * F;
    const unsigned fmagic = (unsigned)((0x100000000ull + (unsigned)Fq - 1) / (unsigned)Fq);   // q / Fq == umulhi(q, fmagic), q < 2^16
    auto frame_of = [&](int q) { return Fq == 1 ? q : (int)__umulhi((unsigned)q, fmagic); };    // (the magic number of 1 does not fit 32 bits)

    // once per workgroup: the folded first layer's fragments -> LDS
    for (int i = tid; i < 3 * KB * 2 * 64; i += kBlock) afr[i] = reinterpret_cast<const uint32x4 *>(d.afrag)[i];
    const float b0[4] = {d.bias0[0], d.bias0[1], d.bias0[2], d.bias0[3]}, w1[4] = {d.w1[0], d.w1[1], d.w1[2], d.w1[3]};
    const double thr = d.thresholds[0];
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(outputs ? outputs + (int64_t)c * E : nullptr, 0, outputs ? (int)(E * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t flg_rs = __builtin_amdgcn_make_buffer_rsrc(flags ? flags + (int64_t)c * E : nullptr, 0, flags ? (int)E : 0, 0x00020000);

    constexpr int NQ = 8;                             // quads per thread: 128 frames x 128 bins / 4 / 512
    // columns of frames e0 .. e0 + 127 are contiguous in memory; quads past the channel's last frame read as zeros
    auto tile_rs = [&](int tile) {
        const int64_t e0 = (int64_t)tile * step;
        int64_t left = tile < tiles_per_channel ? (J - e0) * F : 0;    // floats from the tile's first column to the end of the channel
        left = left < 0 ? 0 : (left > (int64_t)kTile * F ? (int64_t)kTile * F : left);
        if (left > 0) left += 4 * Fq - F;                              // (the last frame's last quad, whole)
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(chan + e0 * F), 0, (int)left * 4, 0x00020000);
    };
    // where this thread's quad k sits in the tile (bytes), and which of its four values are bins of its frame
    int qoff[NQ], qlive[NQ];
#pragma unroll
    for (int k = 0; k < NQ; k++) {
        const int q = tid + kBlock * k, fr = frame_of(q), b4 = q - fr * Fq;
        qoff[k] = q < nq ? 4 * (fr * F + 4 * b4) : 0x7fffff00;          // (past the tile: out of the descriptor's range, zeros)
        qlive[k] = F - 4 * b4;                                          // values 0 .. qlive-1 of the quad are bins
    }
    uint32x4 v[NQ];
    auto load_tile = [&](int tile) {
        const __amdgpu_buffer_rsrc_t rs = tile_rs(tile);
#pragma unroll
        for (int k = 0; k < NQ; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, qoff[k], 0, 0);
    };
    load_tile(blockIdx.x);
    // log / dB columns (SyllableDetector.swift:197-207 vvlogf, :185-195 vDSP_vdbcon with a zero reference of 1: 20 log10 x)
    // through the hardware's base-2 logarithm (1 ulp; a denormal argument is lifted into its range first); ln 0 = -inf as
    // in the reference, which has no guard either
    const float lscale = d.scaling == 1 ? 0.6931471805599453f : 6.020599913279624f;
    auto scaled = [&](float x) {
        const bool tiny = x < 0x1p-96f;
        const float l = __builtin_amdgcn_logf(tiny ? x * 0x1p64f : x) - (tiny ? 64.0f : 0.0f);
        return l * lscale;
    };
    int parity = 0;
    for (int tile = blockIdx.x; tile < tiles_per_channel; tile += gridDim.x, parity ^= 1) {
        const int64_t e0 = (int64_t)tile * step;      // first evaluation = first frame of the tile
        if (d.scaling != 0 || (F & 3) != 0) {
#pragma unroll
            for (int k = 0; k < NQ; k++) {
                floatx4 x = as_floatx4(v[k]);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (d.scaling != 0) x[i] = scaled(x[i]);
                    x[i] = i < qlive[k] ? x[i] : 0.0f;                  // the next frame's bins behind this frame's last ones
                }
                union { floatx4 f; uint32x4 u; } cv;
                cv.f = x;
                v[k] = cv.u;
            }
        }
        // (the sums alternate between two buffers: the previous tile's evaluations may still be reading theirs)
        float *ss = ss0 + parity * kTile;
        // ---- every frame gets its own power-of-two exponent (its column norm goes to [2^12, 2^13)): a quiet frame next to a
        // loud one keeps 22 bits of its own level.  First the frame's sum of squares: per quad, then per frame in a fixed
        // order (no atomics: results do not depend on the order the threads arrive in)
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            const int q = tid + kBlock * k;
            if (q < nq) {
                const floatx4 x = as_floatx4(v[k]);
                pq[q] = fmaf(x[0], x[0], fmaf(x[1], x[1], fmaf(x[2], x[2], x[3] * x[3])));
            }
        }
        __syncthreads();
        if (tid < kTile) {
            float sacc = 0.0f;
            for (int i = 0; i < Fq; i++) sacc += pq[tid * Fq + i];
            ss[tid] = sacc;
            // t = floor(log2 sqrt(ss)) + 64, clamped; up = 2^(76 - t), down = 1 / up (see kernels_fused_r.hip, mag_micro)
            unsigned tb = ((__float_as_uint(sacc) + 0x800000u) >> 1) & 0x7f800000u;
            tb = (unsigned)min(max((int)tb, 16 << 23), 80 << 23);
            fup[tid] = __uint_as_float((203u << 23) - tb);
            fdn[tid] = __uint_as_float(tb + (51u << 23));
        }
        __syncthreads();
        // ---- scale, split into f16 hi + lo, -> LDS [frame][bin]
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            const int q = tid + kBlock * k;
            if (q < nq) {
                const floatx4 x = as_floatx4(v[k]);
                const int fr = frame_of(q), bin = 4 * (q - fr * Fq);
                const float sx = fup[fr];
                unsigned h0, l0, h1, l1;
                split_pair_scaled(x[0], x[1], sx, h0, l0);
                split_pair_scaled(x[2], x[3], sx, h1, l1);
                uint32x2 uh = {h0, h1}, ul = {l0, l1};
                *reinterpret_cast<uint32x2 *>(colh + fr * CS + bin) = uh;
                *reinterpret_cast<uint32x2 *>(coll + fr * CS + bin) = ul;
            }
        }
        // the staging registers are free: the next tile's columns start their way from HBM
        load_tile(tile + (int)gridDim.x);
        // bins 4 Fq .. 32 KB - 1 of every row are read by the last k-block: zeros (the weights there are zero too, but
        // 0 * NaN from stale LDS is not)
        for (int i = tid; i < kTile * ((32 * KB - 4 * Fq) / 4); i += kBlock) {
            const int per = (32 * KB - 4 * Fq) / 4, fr = i / per, bin = 4 * Fq + 4 * (i - fr * per);
            uint32x2 z = {0u, 0u};
            *reinterpret_cast<uint32x2 *>(colh + fr * CS + bin) = z;
            *reinterpret_cast<uint32x2 *>(coll + fr * CS + bin) = z;
        }
        __syncthreads();
        // ---- tap products of this wave's 16 frames: P[(t, h), j] for all taps at once, three row tiles
        {
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
            // result layout: column = frame f, register i of lane group g4 in tile m = row 16 m + 4 g4 + i = tap 4 m + g4,
            // unit i: four consecutive floats of the frame's row in pbuf
            const float dn = fdn[fr];                 // back to true units by the frame's exponent
#pragma unroll
            for (int m = 0; m < 3; m++) *reinterpret_cast<floatx4 *>(pbuf + fr * PS + 4 * (4 * m + g4)) = acc[m] * dn;
        }
        __syncthreads();
        // ---- evaluations e0 .. e0 + step - 1, one thread each: diagonal sum over the taps, l2normalize
        // (NeuralNet.swift:47-59: z and the sums of squares are both in scaled-column units), the rest of the network
        if (tid < step) {
            floatx4 z = {0.f, 0.f, 0.f, 0.f};
            float ssw = 0.0f;
            for (int t = 0; t < T; t++) {
                z += *reinterpret_cast<const floatx4 *>(pbuf + (tid + t) * PS + 4 * t);
                ssw += ss[tid + t];
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
    }
}

}  // namespace

hipError_t launch_mlpx(const MlpxDesc &d, const float *columns, int C, int64_t J, int64_t E, float *outputs, uint8_t *flags,
                       hipStream_t stream)
{
    if (E <= 0 || C <= 0) return hipSuccess;
    // 32-bit byte offsets per channel (outputs)
    if ((uint64_t)E * 4u >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    const int step = kTile - (d.T - 1);
    const int64_t tiles = (E + step - 1) / step;
    // a workgroup keeps the first layer's fragments in LDS and walks tiles of one channel: enough workgroups to fill the
    // chip a few times over
    int64_t per_channel = (4096 + C - 1) / C;
    per_channel = per_channel < 1 ? 1 : (per_channel > tiles ? tiles : per_channel);
    dim3 grid((unsigned)per_channel, (unsigned)C);
    auto launch = [&](auto kern) {
        hipError_t st = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, d.lds_total);
        if (st != hipSuccess) return st;
        hipLaunchKernelGGL(kern, grid, dim3(kBlock), (size_t)d.lds_total, stream, d, columns, J, E, (int)tiles, outputs, flags);
        return hipGetLastError();
    };
    if (d.KB == 1) return launch(mlp_mfma_kernel<1>);
    if (d.KB == 2) return launch(mlp_mfma_kernel<2>);
    if (d.KB == 4) return launch(mlp_mfma_kernel<4>);
    return hipErrorInvalidValue;
}

}  // namespace sd
