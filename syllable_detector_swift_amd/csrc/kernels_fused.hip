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
// MI355X formulation (not a translation of the vDSP call sequence):
//   * the detector only needs F (<= 32) bins of each N-point spectrum, so the windowed DFT of a
//     tile of 32 frames is the GEMM  Xt[2 x 32 rows, 32 frames] = Dt[rows, W] . S[W, 32 frames]
//     with Dt = window o {cos, -sin} and S read straight from the staged sample stream (frame j
//     is just the address j*hop: no per-frame copy, no ring).  It runs on the matrix cores as
//     v_mfma_f32_32x32x16_f16 with every operand split into f16 hi + lo (block floating point,
//     power-of-two scales): hi*hi + hi*lo + lo*hi reproduces an fp32 product to ~2^-21, and the
//     fp32 accumulate keeps the sum; measured error vs the fp64 anchor is below an fp32 FFT's;
//   * the result tile has frames on lanes and bins in registers, which is exactly the B-operand
//     layout of the next MFMA, so the first layer -- folded with the affine input maps into
//     W' = W0 o gain, split per time slot into T*H partial dot products per frame -- runs on the
//     matrix cores too with no data movement;
//   * evaluation e is the diagonal sum  sum_t P[(t,h)][e+t]  over an LDS ring of partials plus a
//     per-window statistic (l2 norm / min-max / mean-std) -- every frame is transformed once,
//     instead of T times as in the reference's sliding re-read.
//
// gfx950 only.  wave = 64; 256-thread workgroups, one per CU (LDS-limited).

#include "kernels.hpp"

namespace sd {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int uint32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;
constexpr int kTile = kFusedTileFrames;

struct __attribute__((packed, aligned(4))) float4_u { float x, y, z, w; };   // 4-byte aligned 16-byte load

__device__ __forceinline__ half8 as_half8(uint32x4 v)
{
    union { uint32x4 u; half8 h; } c;
    c.u = v;
    return c.h;
}

// f32 -> f16 hi + f16 lo with hi = the top 11 significand bits (so the conversion is exact) and
// lo = the exact remainder rounded toward zero: hi + lo == x to 2^-21 relative.
__device__ __forceinline__ void split_pair(float a, float b, unsigned &hi, unsigned &lo)
{
    const float ah = __uint_as_float(__float_as_uint(a) & 0xFFFFE000u);
    const float bh = __uint_as_float(__float_as_uint(b) & 0xFFFFE000u);
    union { decltype(__builtin_amdgcn_cvt_pkrtz(0.f, 0.f)) h; unsigned u; } ch, cl;
    ch.h = __builtin_amdgcn_cvt_pkrtz(ah, bh);
    cl.h = __builtin_amdgcn_cvt_pkrtz(a - ah, b - bh);
    hi = ch.u;
    lo = cl.u;
}

__device__ __forceinline__ float pow2f(int e)   // 2^e for e in [-126, 127]
{
    return __uint_as_float((unsigned)(e + 127) << 23);
}

__device__ __forceinline__ float transfer_fn(int tf, float x)
{
    switch (tf) {
    case 0: return tanhf(x);                         // TanSig  NeuralNet.swift:189-194
    case 1: return 1.0f / (expf(-x) + 1.0f);         // LogSig  :196-215
    case 3: return fminf(fmaxf(x, 0.0f), 1.0f);      // SatLin  :223-228
    default: return x;                               // PureLin :217-221
    }
}

__device__ __forceinline__ floatx16 mfma16(half8 a, half8 b, floatx16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <int KS, int MT>
__global__ void __launch_bounds__(kBlock, 1)
fused_kernel(FusedDesc d, const float *__restrict__ samples, int64_t stride, int64_t S, int64_t E,
             float *__restrict__ outputs, uint8_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32x4 *lds_dfrag = reinterpret_cast<uint32x4 *>(smem + d.lds_dfrag);
    _Float16 *smp_hi = reinterpret_cast<_Float16 *>(smem + d.lds_hi);
    _Float16 *smp_lo = reinterpret_cast<_Float16 *>(smem + d.lds_lo);
    float *pbuf = reinterpret_cast<float *>(smem + d.lds_pbuf);     // rows [0,TH): partials; rows TH, TH+1: window statistics
    float *red = reinterpret_cast<float *>(smem + d.lds_red);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;          // frame column inside the wave's 32-frame tile
    const int hh = lane >> 5;         // lane half: selects k rows 8h..8h+7 of an operand, rows +4 of a result
    const int c = blockIdx.y;
    const int64_t e_b = (int64_t)blockIdx.x * d.seg_evals;
    if (e_b >= E) return;
    const int64_t e_e = (e_b + d.seg_evals < E) ? e_b + d.seg_evals : E;
    const float *row = samples + (int64_t)c * stride;
    const int PS = d.ps;
    const int T = d.T, F = d.F, H = d.H, TH = d.TH;

    // DFT basis fragments -> LDS (64 KB for W = 256), once per workgroup
    for (int i = tid; i < KS * 4 * 64; i += kBlock) lds_dfrag[i] = reinterpret_cast<const uint32x4 *>(d.dfrag)[i];
    // folded first-layer fragments and per-k-step sample offsets -> registers
    half8 wfr[MT][2][2];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int p = 0; p < 2; p++)
                wfr[m][s][p] = as_half8(reinterpret_cast<const uint32x4 *>(d.wfrag)[((m * 2 + s) * 2 + p) * 64 + lane]);
    int ko[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) ko[ks] = d.koff[ks * 2 + hh];
    const int fbase = (32 * wave + r) * (d.hop + d.skew);

    for (int pass = 0; pass < d.runs; pass++) {
        const int64_t jp = e_b + (int64_t)kTile * pass;       // first frame of this pass
        if (jp - (T - 1) >= e_e) break;

        // ---------------- stage: HBM -> registers -> (block max, scale, hi/lo split) -> LDS
        const int64_t g0 = jp * d.hop + d.gap;                 // frame j covers [j*hop + gap, j*hop + gap + W)
        float4 v[kFusedMaxLoads];
        float amax = 0.0f;
#pragma unroll
        for (int k = 0; k < kFusedMaxLoads; k++) {
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < d.nload) {
                const int i = 4 * (tid + kBlock * k);
                const int64_t gi = g0 + i;
                if (i < d.nsmp) {
                    if (gi + 3 < S) {
                        const float4_u t = *reinterpret_cast<const float4_u *>(row + gi);
                        v[k] = make_float4(t.x, t.y, t.z, t.w);
                    } else {
                        if (gi < S) v[k].x = row[gi];
                        if (gi + 1 < S) v[k].y = row[gi + 1];
                        if (gi + 2 < S) v[k].z = row[gi + 2];
                    }
                }
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[k].x), fabsf(v[k].y))), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        // block floating point: scale the tile so its largest sample lands in [2^13, 2^14)
        int se = 13 - (int)((__float_as_uint(amax) >> 23) & 0xffu) + 127;   // exponent of the scale
        se = amax > 0.0f ? (se < -100 ? -100 : (se > 100 ? 100 : se)) : 0;
        const float sx = pow2f(se);
#pragma unroll
        for (int k = 0; k < kFusedMaxLoads; k++) {
            if (k < d.nload) {
                const int i = 4 * (tid + kBlock * k);
                if (i < d.nsmp) {
                    unsigned h0, l0, h1, l1;
                    split_pair(v[k].x * sx, v[k].y * sx, h0, l0);
                    split_pair(v[k].z * sx, v[k].w * sx, h1, l1);
                    const int p = i + d.skew * (int)__umulhi((unsigned)i, d.hop_magic);
                    *reinterpret_cast<uint2 *>(smp_hi + p) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2 *>(smp_lo + p) = make_uint2(l0, l1);
                }
            }
        }
        __syncthreads();

        // ---------------- band-limited DFT of this wave's 32 frames on the matrix cores
        floatx16 acc_re = {0}, acc_im = {0};
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const _Float16 *ph = smp_hi + fbase + ko[ks];
            const _Float16 *pl = smp_lo + fbase + ko[ks];
            const half4 bh0 = *reinterpret_cast<const half4 *>(ph), bh1 = *reinterpret_cast<const half4 *>(ph + 4);
            const half4 bl0 = *reinterpret_cast<const half4 *>(pl), bl1 = *reinterpret_cast<const half4 *>(pl + 4);
            const half8 bh = __builtin_shufflevector(bh0, bh1, 0, 1, 2, 3, 4, 5, 6, 7);
            const half8 bl = __builtin_shufflevector(bl0, bl1, 0, 1, 2, 3, 4, 5, 6, 7);
            const half8 a_re_h = as_half8(lds_dfrag[((ks * 2 + 0) * 2 + 0) * 64 + lane]);
            const half8 a_re_l = as_half8(lds_dfrag[((ks * 2 + 0) * 2 + 1) * 64 + lane]);
            const half8 a_im_h = as_half8(lds_dfrag[((ks * 2 + 1) * 2 + 0) * 64 + lane]);
            const half8 a_im_l = as_half8(lds_dfrag[((ks * 2 + 1) * 2 + 1) * 64 + lane]);
            acc_re = mfma16(a_re_h, bh, acc_re);
            acc_im = mfma16(a_im_h, bh, acc_im);
            acc_re = mfma16(a_re_h, bl, acc_re);
            acc_im = mfma16(a_im_h, bl, acc_im);
            acc_re = mfma16(a_re_l, bh, acc_re);
            acc_im = mfma16(a_im_l, bh, acc_im);
        }

        // ---------------- magnitude (zvabs/2 :329-333 or zvmags/4 :270-274), scaling
        // (SyllableDetector.swift:184-212), per-frame statistic, f16 split for the next MFMA.
        // Result layout: column = frame r, register g of lane half hh = bin row (g&3) + 8(g>>2) + 4hh.
        const float inv = pow2f(-se - 13);                   // accumulators hold X * sx * 2^13
        // scale that keeps the column inside f16 range for the layer-0 MFMA (exact power of two)
        float cs = 1.0f;
        if (d.scaling == 0) cs = d.power_mode ? pow2f(2 * (se < 40 ? (se > -40 ? se : -40) : 40) - 30) : pow2f(se - 8);
        float cval[16];
        float st0 = 0.0f, st1 = 0.0f;
        if (d.norm == 2) { st0 = INFINITY; st1 = -INFINITY; }
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const float re = acc_re[g] * inv, im = acc_im[g] * inv;
            const float pw = re * re + im * im;
            float x = d.power_mode ? pw : sqrtf(pw);
            if (d.scaling == 1) x = logf(x);
            else if (d.scaling == 2) x = 20.0f * log10f(x);
            const bool valid = ((g & 3) + 8 * (g >> 2) + 4 * hh) < F;
            x = valid ? x : 0.0f;
            cval[g] = x;
            if (d.norm == 1) st0 = fmaf(x, x, st0);
            else if (d.norm == 2) { st0 = valid ? fminf(st0, x) : st0; st1 = valid ? fmaxf(st1, x) : st1; }
            else if (d.norm == 3) st0 += x;
        }
        if (d.norm == 1) {
            st0 += __shfl_xor(st0, 32, 64);
        } else if (d.norm == 2) {
            st0 = fminf(st0, __shfl_xor(st0, 32, 64));
            st1 = fmaxf(st1, __shfl_xor(st1, 32, 64));
        } else if (d.norm == 3) {
            st0 += __shfl_xor(st0, 32, 64);
            st0 = st0 / (float)F;                            // mean of this frame's column
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const bool valid = ((g & 3) + 8 * (g >> 2) + 4 * hh) < F;
                const float dlt = cval[g] - st0;
                st1 = valid ? fmaf(dlt, dlt, st1) : st1;     // M2 of this frame's column
            }
            st1 += __shfl_xor(st1, 32, 64);
        }
        const int slot = (T - 1) + 32 * wave + r;
        if (hh == 0 && d.norm != 0) {
            pbuf[TH * PS + slot] = st0;
            pbuf[(TH + 1) * PS + slot] = st1;
        }
        half8 bh2[2], bl2[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            uint32x4 uh, ul;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                unsigned a, b;
                split_pair(cval[8 * s + 2 * j] * cs, cval[8 * s + 2 * j + 1] * cs, a, b);
                uh[j] = a;
                ul[j] = b;
            }
            bh2[s] = as_half8(uh);
            bl2[s] = as_half8(ul);
        }

        // ---------------- first layer, folded and split per time slot: P[(t,h)][frame] on the matrix cores
        const float unscale = d.w_unscale / cs;
#pragma unroll
        for (int m = 0; m < MT; m++) {
            floatx16 pacc = {0};
#pragma unroll
            for (int s = 0; s < 2; s++) {
                pacc = mfma16(wfr[m][s][0], bh2[s], pacc);
                pacc = mfma16(wfr[m][s][0], bl2[s], pacc);
                pacc = mfma16(wfr[m][s][1], bh2[s], pacc);
            }
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const int prow = 32 * m + (g & 3) + 8 * (g >> 2) + 4 * hh;
                if (prow < TH) pbuf[prow * PS + slot] = pacc[g] * unscale;
            }
        }
        __syncthreads();

        // ---------------- evaluations completed by this pass: e = jp - (T-1) + q uses slots q..q+T-1
        {
            const int q = 32 * wave + r;
            const int64_t e = jp - (T - 1) + q;
            const bool valid = e >= e_b && e < e_e;
            float alpha = 1.0f, beta = 0.0f;
            const float *s0 = pbuf + TH * PS + q, *s1 = pbuf + (TH + 1) * PS + q;
            if (d.norm == 1) {                                // L2Normalize, NeuralNet.swift:47-59
                float ssw = 0.0f;
                for (int t = 0; t < T; t++) ssw += s0[t];
                alpha = 1.0f / sqrtf(ssw);
            } else if (d.norm == 2) {                         // Normalize, :69-96
                float mn = INFINITY, mx = -INFINITY;
                for (int t = 0; t < T; t++) { mn = fminf(mn, s0[t]); mx = fmaxf(mx, s1[t]); }
                const float range = mx - mn;
                if (range == 0.0f) { alpha = 0.0f; beta = -1.0f; }
                else { alpha = 2.0f / range; beta = (0.0f - mn - mx) / range; }
            } else if (d.norm == 3) {                         // NormalizeStd, :105-108 (population sigma)
                float n = 0.0f, mean = 0.0f, m2 = 0.0f;
                for (int t = 0; t < T; t++) {                 // pairwise-stable combination of per-frame (mean, M2)
                    const float nb = (float)F, tot = n + nb, dlt = s0[t] - mean;
                    mean += dlt * nb / tot;
                    m2 += s1[t] + dlt * dlt * n * nb / tot;
                    n = tot;
                }
                const float sd = sqrtf(m2 / (float)d.I);
                alpha = 1.0f / sd;
                beta = -mean / sd;
            }
            // hidden unit h of lane half hh: h = 2*i + hh
            float act[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int h = 2 * i + hh;
                act[i] = 0.0f;
                if (h < H) {
                    float z = 0.0f;
                    for (int t = 0; t < T; t++) z += pbuf[(t * H + h) * PS + q + t];
                    act[i] = transfer_fn(d.tf0, fmaf(alpha, z, fmaf(beta, d.rvec[h], d.bias0[h])));
                }
            }
            const int64_t obase = ((int64_t)c * E + e) * d.n_out;
            if (d.n_layers == 2) {
                bool hit = false;
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    if (o < d.n_out) {
                        float y = 0.0f;
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const int h = 2 * i + hh;
                            if (h < H) y = fmaf(d.w1[o * H + h], act[i], y);
                        }
                        y += __shfl_xor(y, 32, 64);
                        y = transfer_fn(d.tf1, y + d.b1[o]);
                        for (int k = 0; k < d.n_out_fns; k++) {       // reverse maps, NeuralNet.swift:137-142 / :175-180
                            const float *op = d.out_params + k * (1 + 2 * d.n_out);
                            y = (y - op[0]) / op[1 + o] + op[1 + d.n_out + o];
                        }
                        if (valid && hh == 0 && outputs) outputs[obase + o] = y;
                        if (o == 0 || d.rule == 1) hit = hit || ((double)y >= d.thresholds[o]);
                    }
                }
                if (valid && hh == 0 && flags) flags[(int64_t)c * E + e] = hit ? 1 : 0;
            } else {
                bool hit = false;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int h = 2 * i + hh;
                    if (h < H) {
                        float y = act[i];
                        for (int k = 0; k < d.n_out_fns; k++) {
                            const float *op = d.out_params + k * (1 + 2 * d.n_out);
                            y = (y - op[0]) / op[1 + h] + op[1 + d.n_out + h];
                        }
                        if (valid && outputs) outputs[obase + h] = y;
                        if (h == 0 || d.rule == 1) hit = hit || ((double)y >= d.thresholds[h]);
                    }
                }
                const bool other = __shfl_xor((int)hit, 32, 64) != 0;
                if (valid && hh == 0 && flags) flags[(int64_t)c * E + e] = (hit || other) ? 1 : 0;
            }
        }
        __syncthreads();
        // ---------------- keep the last T-1 frames' partials and statistics for the next pass
        for (int idx = tid; idx < (TH + 2) * (T - 1); idx += kBlock) {
            const int prow = idx / (T - 1), t = idx - prow * (T - 1);
            pbuf[prow * PS + t] = pbuf[prow * PS + kTile + t];
        }
        __syncthreads();
    }
}

template <int KS, int MT>
hipError_t launch_one(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t S, int64_t E,
                      float *outputs, uint8_t *flags, hipStream_t stream)
{
    auto kern = fused_kernel<KS, MT>;
    hipError_t st = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, d.lds_total);
    if (st != hipSuccess) return st;
    const int64_t segs = (E + d.seg_evals - 1) / d.seg_evals;
    dim3 grid((unsigned)segs, (unsigned)C);
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), (size_t)d.lds_total, stream, d, samples, stride, S, E, outputs, flags);
    return hipGetLastError();
}

}  // namespace

int fused_supported(int KS, int MT)
{
    return (KS == 16 || KS == 8) && (MT == 1 || MT == 2 || MT == 4) ? 1 : 0;
}

hipError_t launch_fused(const FusedDesc &d, const float *samples, int64_t stride, int C, int64_t S, int64_t J,
                        int64_t E, float *outputs, uint8_t *flags, hipStream_t stream)
{
    (void)J;
    if (E <= 0 || C <= 0) return hipSuccess;
#define SD_CASE(K, M) \
    if (d.KS == K && d.MT == M) return launch_one<K, M>(d, samples, stride, C, S, E, outputs, flags, stream)
    SD_CASE(16, 1);
    SD_CASE(16, 2);
    SD_CASE(16, 4);
    SD_CASE(8, 1);
    SD_CASE(8, 2);
    SD_CASE(8, 4);
#undef SD_CASE
    return hipErrorInvalidValue;
}

}  // namespace sd
