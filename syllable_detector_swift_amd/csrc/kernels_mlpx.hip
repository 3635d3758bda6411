// kernels_mlpx.hip -- the network stage of the generic engine on the matrix cores.
//
// Reference path being replaced (reference root relative): SyllableDetector.processNewValue's window of timeRange
// columns (Common/SyllableDetector.swift:153-217) -> NeuralNet.apply (Common/NeuralNet.swift:294-326, :366-377;
// L2Normalize :47-59, MapMinMax / MapStd :127-131 / :162-169 and their reverse maps, TanSig :189-194) -> lastDetected
// (Common/SyllableDetector.swift:27-31), for spectrogram columns that are already in HBM ([C][J][F] fp32, written by the
// generic STFT kernels: long windows, wide bands -- BASELINE configs[2]).
//
// The fused engine's formulation, fed from memory instead of from its own DFT: a tile of 128 evaluations reads its
// 128 + timeRange - 1 columns once (coalesced quads), scales them by one power of two (the tile's largest value goes to
// [2^13, 2^14): block floating point), splits them into f16 hi + lo in LDS, and the first layer -- folded on the host with
// the affine input maps -- is a GEMM on the matrix cores whose B operand for tap t is simply the column buffer at row
// offset e + t (v_mfma_f32_16x16x32_f16; hi*hi + hi*lo + lo*hi reproduces the fp32 product, fp32 accumulate).  The
// l2normalize denominator comes from per-frame sums of squares accumulated while the columns are split (64-bit fixed
// point in LDS, so that the sum does not depend on the order the threads arrive in).  The rest of the network runs in
// registers.  The interpretive kernels (kernels_generic.hip) evaluate the same network with one wave per evaluation on
// the vector units: 2.7 ms against this kernel's time on the configs[2] batch.
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
    unsigned long long *ss = reinterpret_cast<unsigned long long *>(smem + d.lds_ss);
    float *red = reinterpret_cast<float *>(smem + d.lds_red);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = lane & 15, g4 = lane >> 4;
    const int c = blockIdx.y;
    const int F = d.F, T = d.T, CS = d.col_stride;
    const int nf = kTile + T - 1;                     // frames a tile's windows cover
    const int nq = nf * F / 4;                        // quads of column values per tile (F is a multiple of 4)
    const float *chan = columns + (int64_t)c * J * F;
    const unsigned fmagic = (unsigned)((0x100000000ull + (unsigned)(F / 4) - 1) / (unsigned)(F / 4));   // q / (F/4) == umulhi(q, fmagic), q < 2^16

    // once per workgroup: the folded first layer's fragments -> LDS
    for (int i = tid; i < T * KB * 2 * 64; i += kBlock) afr[i] = reinterpret_cast<const uint32x4 *>(d.afrag)[i];
    float c_b0[4], c_w1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {                     // rows 4*g4 + j of a result: only lane group 0 holds hidden units (H <= 4)
        c_b0[j] = g4 == 0 ? d.bias0[j] : 0.0f;
        c_w1[j] = g4 == 0 ? d.w1[j] : 0.0f;
    }
    const double thr = d.thresholds[0];
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(outputs ? outputs + (int64_t)c * E : nullptr, 0, outputs ? (int)(E * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t flg_rs = __builtin_amdgcn_make_buffer_rsrc(flags ? flags + (int64_t)c * E : nullptr, 0, flags ? (int)E : 0, 0x00020000);

    constexpr int NQ = 9;                             // quads per thread: (128 + 11) * 128 / 4 / 512 rounded up
    for (int tile = blockIdx.x; tile < tiles_per_channel; tile += gridDim.x) {
        const int64_t e0 = (int64_t)tile * kTile;     // first evaluation = first frame of the tile
        // ---- columns of frames e0 .. e0 + nf - 1: contiguous in memory; quads past the channel's last frame read as zeros
        const int64_t left = (J - e0) * F;            // floats from the tile's first column to the end of the channel
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(chan + e0 * F), 0, (int)((left < (int64_t)nf * F ? left : (int64_t)nf * F) * 4), 0x00020000);
        uint32x4 v[NQ];
#pragma unroll
        for (int k = 0; k < NQ; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * (tid + kBlock * k), 0, 0);
        if (tid < nf) ss[tid] = 0ull;
        float amax = 0.0f;
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            const floatx4 q = as_floatx4(v[k]);
            amax = absmax3(absmax3(amax, q[0], q[1]), q[2], q[3]);
        }
        amax = wave_max_nonneg(amax);
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        int se;
        {
            const floatx4 r0 = *reinterpret_cast<const floatx4 *>(red), r1 = *reinterpret_cast<const floatx4 *>(red + 4);
            const float m = fmaxf(fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[2], r0[3])), fmaxf(fmaxf(r1[0], r1[1]), fmaxf(r1[2], r1[3])));
            int e = 13 - (int)((__float_as_uint(m) >> 23) & 0xffu) + 127;
            e = m > 0.0f ? (e < -100 ? -100 : (e > 100 ? 100 : e)) : 0;
            se = __builtin_amdgcn_readfirstlane(e);
        }
        const float sx = pow2f(se);
        // ---- scale, split into f16 hi + lo, -> LDS [frame][bin]; the frame's sum of squares (of the scaled values)
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            const int q = tid + kBlock * k;
            if (q < nq) {
                const floatx4 x = as_floatx4(v[k]);
                const int fr = F == 4 ? q : (int)__umulhi((unsigned)q, fmagic), bin = 4 * (q - fr * (F / 4));   // (the magic number of 1 does not fit 32 bits)
                unsigned h0, l0, h1, l1;
                split_pair_scaled(x[0], x[1], sx, h0, l0);
                split_pair_scaled(x[2], x[3], sx, h1, l1);
                uint32x2 uh = {h0, h1}, ul = {l0, l1};
                *reinterpret_cast<uint32x2 *>(colh + fr * CS + bin) = uh;
                *reinterpret_cast<uint32x2 *>(coll + fr * CS + bin) = ul;
                const float a0 = x[0] * sx, a1 = x[1] * sx, a2 = x[2] * sx, a3 = x[3] * sx;
                const float p = fmaf(a0, a0, fmaf(a1, a1, fmaf(a2, a2, a3 * a3)));      // < 2^30
                // p * 2^20 as a 64-bit integer (integer sums do not depend on the order the threads arrive in):
                // p = ph * 4096 + rem exactly, so p * 2^20 = ph * 2^32 + rem * 2^20 with rem * 2^20 < 2^32
                const unsigned ph = (unsigned)(p * (1.0f / 4096.0f));
                const unsigned pl = (unsigned)(fmaf(-(float)ph, 4096.0f, p) * 1048576.0f);
                atomicAdd(&ss[fr], ((unsigned long long)ph << 32) | pl);
            }
        }
        // bins F .. 32 KB - 1 of every row are read by the last k-block: zero them once per tile (the weights there are zero
        // too, but 0 * NaN from stale LDS is not)
        for (int i = tid; i < nf * ((32 * KB - F) / 4); i += kBlock) {
            const int per = (32 * KB - F) / 4, fr = i / per, bin = F + 4 * (i - fr * per);
            uint32x2 z = {0u, 0u};
            *reinterpret_cast<uint32x2 *>(colh + fr * CS + bin) = z;
            *reinterpret_cast<uint32x2 *>(coll + fr * CS + bin) = z;
        }
        __syncthreads();
        // ---- first layer: Z[h, e] = sum_t sum_kb W'_{t,kb}[h, :] . C[32 kb .., e + t]; this wave's 16 evaluations
        const int slot = 16 * wave + f;
        floatx4 z = {0.f, 0.f, 0.f, 0.f}, z2 = {0.f, 0.f, 0.f, 0.f};
        const _Float16 *bph = colh + slot * CS + 8 * g4, *bpl = coll + slot * CS + 8 * g4;
        for (int t = 0; t < T; t++) {
#pragma unroll
            for (int kb = 0; kb < KB; kb++) {
                const half8 bh = as_half8(*reinterpret_cast<const uint32x4 *>(bph + t * CS + 32 * kb));
                const half8 bl = as_half8(*reinterpret_cast<const uint32x4 *>(bpl + t * CS + 32 * kb));
                const half8 ah = as_half8(afr[((t * KB + kb) * 2 + 0) * 64 + lane]), al = as_half8(afr[((t * KB + kb) * 2 + 1) * 64 + lane]);
                z = mfma(ah, bh, z);
                z2 = mfma(ah, bl, z2);
                z2 = mfma(al, bh, z2);
            }
        }
        z += z2;
        // ---- l2normalize (NeuralNet.swift:47-59): z and the sums of squares are both in scaled-column units
        unsigned long long ssw = 0ull;
        for (int t = 0; t < T; t++) ssw += ss[slot + t];
        const float alpha = d.w_unscale * __builtin_amdgcn_rsqf((float)ssw * (1.0f / 1048576.0f));
        float y = d.b1;
#pragma unroll
        for (int j = 0; j < 4; j++) {                 // TanSig hidden units (padding rows meet zero weights), linear output
            const float a = fmaf(alpha, z[j], c_b0[j]);
            const float th = fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a * 2.885390081777927f) + 1.0f), 1.0f);
            y = fmaf(c_w1[j], th, y);
        }
        y = (y - d.oa) / d.og + d.ob;
        const int64_t e = e0 + slot;
        const bool st = g4 == 0 && e < E;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), out_rs, st ? (unsigned)e * 4u : 0xFFFFFFFFu, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)((double)y >= thr ? 1 : 0), flg_rs, st ? (unsigned)e : 0xFFFFFFFFu, 0, 0);
        __syncthreads();          // the column buffer and the sums are free for the next tile
    }
}

}  // namespace

hipError_t launch_mlpx(const MlpxDesc &d, const float *columns, int C, int64_t J, int64_t E, float *outputs, uint8_t *flags,
                       hipStream_t stream)
{
    if (E <= 0 || C <= 0) return hipSuccess;
    // 32-bit byte offsets per channel (outputs) and per tile (columns)
    if ((uint64_t)E * 4u >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    const int64_t tiles = (E + kTile - 1) / kTile;
    // a workgroup keeps the first layer's fragments in LDS and walks tiles of one channel: enough workgroups to fill the
    // chip a few times over, few enough that loading the fragments stays a small part of the work
    int64_t per_channel = (4096 + C - 1) / C;
    per_channel = per_channel < 1 ? 1 : (per_channel > tiles ? tiles : per_channel);
    dim3 grid((unsigned)per_channel, (unsigned)C);
    auto launch = [&](auto kern) {
        hipError_t st = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, d.lds_total);
        if (st != hipSuccess) return st;
        hipLaunchKernelGGL(kern, grid, dim3(kBlock), (size_t)d.lds_total, stream, d, columns, J, E, (int)tiles, outputs, flags);
        return hipGetLastError();
    };
    if (d.KB == 2) return launch(mlp_mfma_kernel<2>);
    if (d.KB == 4) return launch(mlp_mfma_kernel<4>);
    return hipErrorInvalidValue;
}

}  // namespace sd
