// kernels_aux.hip -- the two data-movement steps either side of the hot path:
//   resample_linear_kernel   ResamplerLinear.resampleVector, Common/Resampler.swift:36-69
//   deinterleave_kernel      appendInterleavedData's strided copy, Common/CircularShortTimeFourierTransform.swift:203-217,
//                            for all channels of a frame-major buffer at once
// Both are HBM-bound gathers: no matrix work, coalesced on the wide side, one pass over the data.
//
// gfx950 only.  wave = 64.

#include "kernels.hpp"

// bit-exact float bookkeeping below: this file is compiled with -ffp-contract=off (see the Makefile), every
// multiply and add rounds on its own exactly as in the reference's vDSP calls

namespace sd {

namespace {

// One output sample per thread.  The arithmetic is the reference's, operation by operation and without
// contraction (so that results are bit-identical to the restatement in oracle/):
//   index  b = offset + float(i) * step                      vDSP_vramp   :52
//          b = 0 for i == 0 when the previous call left a negative offset       :54-56
//   value  a[k] + (b - k) * (a[k+1] - a[k]),  k = floor(b)   vDSP_vlint   :59
//          out[0] = last * (0 - offset) + data[0] * (1 + offset) in that case   :61-63
//   carry  last = data[n_in - 1]                                                :66
// (the new offset depends on sizes only and is computed by the host, see syldet_resample_device)
__global__ void __launch_bounds__(256)
resample_linear_kernel(const float *__restrict__ in, int64_t n_in, int64_t in_stride, float *__restrict__ out,
                       int64_t n_out, int64_t out_stride, float step, float offset, float *__restrict__ last)
{
    const int c = blockIdx.y;
    const float *a = in + (int64_t)c * in_stride;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const bool across = offset < 0.0f;
    float b = offset + (float)i * step;
    if (i == 0 && across) b = 0.0f;
    int64_t k = (int64_t)floorf(b);
    const float frac = b - (float)k;                // (of the unclamped position, as vDSP_vlint computes it)
    k = k < n_in - 1 ? k : n_in - 1;                // a ramp rounded up to n_in must not read past the buffer
    const float a0 = a[k], a1 = (k + 1 < n_in) ? a[k + 1] : a0;
    float y = a0 + frac * (a1 - a0);
    if (i == 0) {
        if (across) y = (last[c] * (0.0f - offset)) + (a[0] * (1.0f + offset));
        last[c] = a[n_in - 1];
    }
    out[(int64_t)c * out_stride + i] = y;
}

// Frame-major [n_frames][total] -> channel-major rows for channels first .. first + C - 1.
// A workgroup moves a tile of 256 frames x 32 channels through LDS: reads run along the interleaved
// buffer (consecutive lanes = consecutive channels of a frame), writes along each channel's row.
constexpr int kTileFrames = 256, kTileCh = 32;

__global__ void __launch_bounds__(256)
deinterleave_kernel(const float *__restrict__ in, int64_t n_frames, int total, int first, int C, float *__restrict__ out,
                    int64_t out_stride)
{
    __shared__ float tile[kTileFrames][kTileCh + 1];
    const int64_t f0 = (int64_t)blockIdx.x * kTileFrames;
    const int c0 = blockIdx.y * kTileCh;
    const int nc = min(kTileCh, C - c0);
    const int tid = threadIdx.x;
    // load: element e of the tile = (frame e / nc, channel e % nc)
    const int nf = (int)min((int64_t)kTileFrames, n_frames - f0);
    for (int e = tid; e < nf * nc; e += 256) {
        const int fr = e / nc, ch = e - fr * nc;
        tile[fr][ch] = in[(f0 + fr) * (int64_t)total + first + c0 + ch];
    }
    __syncthreads();
    for (int ch = 0; ch < nc; ch++)
        if (tid < nf) out[(int64_t)(c0 + ch) * out_stride + f0 + tid] = tile[tid][ch];
}

// Whole-recording rate conversion for offline input: output sample i reads position i * rate_in / rate_out, computed in
// fp64 (an fp32 ramp loses half a sample after ~95 s at 44.1 kHz), linear interpolation between the two neighbours.
// Stateless: no carry, no buffer seams.
__global__ void __launch_bounds__(256)
convert_rate_kernel(const float *__restrict__ in, int64_t n_in, int64_t in_stride, float *__restrict__ out, int64_t n_out,
                    int64_t out_stride, double step)
{
    const int c = blockIdx.y;
    const float *a = in + (int64_t)c * in_stride;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const double pos = (double)i * step;
    int64_t k = (int64_t)pos;
    k = k < n_in - 1 ? k : n_in - 1;
    const double frac = pos - (double)k;
    const double a0 = (double)a[k], a1 = (double)a[k + 1 < n_in ? k + 1 : k];
    out[(int64_t)c * out_stride + i] = (float)(a0 + frac * (a1 - a0));
}

}  // namespace

hipError_t launch_convert_rate(const float *in, int64_t n_in, int64_t in_stride, float *out, int64_t n_out, int64_t out_stride,
                               int C, double step, hipStream_t stream)
{
    if (n_out <= 0 || C <= 0) return hipSuccess;
    dim3 grid((unsigned)((n_out + 255) / 256), (unsigned)C);
    hipLaunchKernelGGL(convert_rate_kernel, grid, dim3(256), 0, stream, in, n_in, in_stride, out, n_out, out_stride, step);
    return hipGetLastError();
}

hipError_t launch_resample_linear(const float *in, int64_t n_in, int64_t in_stride, float *out, int64_t n_out,
                                  int64_t out_stride, int C, float step, float offset, float *last, hipStream_t stream)
{
    if (n_out <= 0 || C <= 0) return hipSuccess;
    dim3 grid((unsigned)((n_out + 255) / 256), (unsigned)C);
    hipLaunchKernelGGL(resample_linear_kernel, grid, dim3(256), 0, stream, in, n_in, in_stride, out, n_out, out_stride, step,
                       offset, last);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Detection flags as bits, for the one exchange of the multi-GPU path (the gather of the
// `uint8 [C/G][E]` flags is latency- and wire-bound; packed it is an eighth of the bytes).
// bits[row][t] bit b = flags[row][8 t + b] != 0; rows are padded to whole bytes.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pack_flags_kernel(const uint8_t *__restrict__ flags, int64_t row_len, int64_t row_bytes, uint8_t *__restrict__ bits)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= row_bytes) return;
    const uint8_t *src = flags + (int64_t)blockIdx.y * row_len + 8 * t;
    const int64_t left = row_len - 8 * t;
    unsigned b = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) b |= (k < left && src[k] != 0) ? (1u << k) : 0u;
    bits[(int64_t)blockIdx.y * row_bytes + t] = (uint8_t)b;
}

// Source row of output row r when the bit rows arrive as `shards` padded blocks of `padded` rows each (the all-gather of
// ragged contiguous channel blocks: the first `extra` blocks hold base + 1 rows, the rest base; padded >= base + 1).  With
// extra == 0 and padded == base the map is the identity.
__device__ __forceinline__ int64_t gathered_row(int64_t r, int64_t padded, int64_t base, int64_t extra)
{
    // (rows <= 65535: 32-bit divisions)
    const unsigned r32 = (unsigned)r, b32 = (unsigned)base, big = (unsigned)extra * (b32 + 1);
    if (r32 < big) return (int64_t)(r32 / (b32 + 1)) * padded + r32 % (b32 + 1);
    const unsigned q = r32 - big;
    return (extra + (int64_t)(q / b32)) * padded + q % b32;
}

// MODE 0: bit row r is row r of `bits`; 1: the gathered layout above behind ONE base pointer (the all-gather's receive buffer); 2: the
// same layout with every block behind a pointer of its own (`from`: a shard's rows are read where they lie -- its send buffer on
// this device, or the slot a peer copy filled -- the copy exchange of the one-process bank makes no copy it does not need)
template <int MODE>
__global__ void __launch_bounds__(256)
unpack_flags_kernel(const uint8_t *__restrict__ bits, int64_t rows, int64_t row_len, int64_t row_bytes, uint8_t *__restrict__ flags,
                    int64_t padded, int64_t base, int64_t extra, const FlagSources from)
{
    constexpr bool MAPPED = MODE != 0;
    // the first byte of source bit row `src` (a row of the gathered layout: block src / padded, row src % padded of it)
    auto row_ptr = [&](int64_t src) -> const uint8_t * {
        if (MODE != 2) return bits + src * row_bytes;
        const unsigned blk = (unsigned)src / (unsigned)padded;
        return from.p[blk] + (int64_t)((unsigned)src - blk * (unsigned)padded) * row_bytes;
    };
    // one aligned 8-byte store per thread over the flat [rows * row_len] output; a thread's eight flags may straddle two rows
    const int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8, total = rows * row_len;
    if (q >= total) return;
    // (the plane's row of flat index q: a 32-bit division wherever the plane is under 2^32 flags -- 4096 x 15 877 is 6.5e7)
    int64_t row, i;
    if (total < (int64_t)0xffffffffLL) {
        const unsigned r32 = (unsigned)q / (unsigned)row_len;
        row = r32;
        i = (int64_t)((unsigned)q - r32 * (unsigned)row_len);
    } else {
        row = q / row_len;
        i = q - row * row_len;
    }
    if (i + 8 <= row_len) {
        // eight flags of ONE row (all but a row's last group): bits i .. i + 7 of it sit in two bytes at most; bit k -> byte k by one
        // multiplication (round 5: the byte-a-bit loop below took 35 us for 512 x 15 877 flags, and on eight GPUs every device
        // unpacks eight times that per batch)
        const int64_t src = MAPPED ? gathered_row(row, padded, base, extra) : row;
        const uint8_t *p = row_ptr(src) + (i >> 3);
        unsigned w = p[0];
        if ((i & 7) != 0) w |= (unsigned)p[1] << 8;               // (i + 8 <= row_len and i % 8 != 0: the next byte is inside the row)
        const uint64_t b = (w >> (i & 7)) & 0xffu;
        const uint64_t spread = (b * 0x0101010101010101ULL) & 0x8040201008040201ULL;      // byte k keeps bit k
        const uint64_t out8 = ((spread + 0x7f7f7f7f7f7f7f7fULL) >> 7) & 0x0101010101010101ULL;   // non-zero byte -> 1
        *reinterpret_cast<uint64_t *>(flags + q) = out8;
        return;
    }
    uint64_t out = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (q + k < total) {
            const int64_t src = MAPPED ? gathered_row(row, padded, base, extra) : row;
            const unsigned b = row_ptr(src)[i >> 3];
            out |= (uint64_t)((b >> (i & 7)) & 1u) << (8 * k);
            if (++i == row_len) { i = 0; row++; }
        }
    }
    if (q + 8 <= total) {
        *reinterpret_cast<uint64_t *>(flags + q) = out;          // hipMalloc'ed bases are 256-byte aligned; q is a multiple of 8
    } else {
        for (int k = 0; q + k < total; k++) flags[q + k] = (uint8_t)(out >> (8 * k));
    }
}

hipError_t launch_pack_flags(const uint8_t *flags, int64_t rows, int64_t row_len, uint8_t *bits, hipStream_t stream)
{
    if (rows <= 0 || row_len <= 0) return hipSuccess;
    const int64_t row_bytes = (row_len + 7) / 8;
    dim3 grid((unsigned)((row_bytes + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL(pack_flags_kernel, grid, dim3(256), 0, stream, flags, row_len, row_bytes, bits);
    return hipGetLastError();
}

hipError_t launch_unpack_flags(const uint8_t *bits, int64_t rows, int64_t row_len, uint8_t *flags, hipStream_t stream)
{
    if (rows <= 0 || row_len <= 0) return hipSuccess;
    const int64_t row_bytes = (row_len + 7) / 8;
    const int64_t threads = (rows * row_len + 7) / 8;
    const bool aligned = (reinterpret_cast<uintptr_t>(flags) & 7) == 0;
    if (!aligned || threads > 0x7fffffffLL * 256) return hipErrorInvalidValue;
    dim3 grid((unsigned)((threads + 255) / 256));
    hipLaunchKernelGGL(unpack_flags_kernel<0>, grid, dim3(256), 0, stream, bits, rows, row_len, row_bytes, flags, (int64_t)0, (int64_t)1, (int64_t)0, FlagSources{});
    return hipGetLastError();
}

// The gathered form: `shards` blocks of `padded` bit rows each, of which block s holds rows/shards (+ 1 for the first
// rows % shards blocks) real rows -- dist.shard_channels' table -- unpacked into the contiguous [rows][row_len] flags.
hipError_t launch_unpack_flags_gathered(const uint8_t *bits, int64_t rows, int64_t row_len, int64_t shards, int64_t padded,
                                        uint8_t *flags, hipStream_t stream)
{
    if (rows <= 0 || row_len <= 0) return hipSuccess;
    if (shards <= 0 || rows < shards || padded < (rows + shards - 1) / shards) return hipErrorInvalidValue;
    const int64_t row_bytes = (row_len + 7) / 8;
    const int64_t threads = (rows * row_len + 7) / 8;
    const bool aligned = (reinterpret_cast<uintptr_t>(flags) & 7) == 0;
    if (!aligned || threads > 0x7fffffffLL * 256) return hipErrorInvalidValue;
    dim3 grid((unsigned)((threads + 255) / 256));
    hipLaunchKernelGGL(unpack_flags_kernel<1>, grid, dim3(256), 0, stream, bits, rows, row_len, row_bytes, flags, padded, rows / shards, rows % shards, FlagSources{});
    return hipGetLastError();
}

// ... with block s at from.p[s] (shards <= kMaxFlagSources)
hipError_t launch_unpack_flags_from(const FlagSources &from, int64_t rows, int64_t row_len, int64_t shards, int64_t padded,
                                    uint8_t *flags, hipStream_t stream)
{
    if (rows <= 0 || row_len <= 0) return hipSuccess;
    if (shards <= 0 || shards > kMaxFlagSources || rows < shards || padded < (rows + shards - 1) / shards) return hipErrorInvalidValue;
    for (int64_t i = 0; i < shards; i++)
        if (!from.p[i]) return hipErrorInvalidValue;
    const int64_t row_bytes = (row_len + 7) / 8;
    const int64_t threads = (rows * row_len + 7) / 8;
    const bool aligned = (reinterpret_cast<uintptr_t>(flags) & 7) == 0;
    if (!aligned || threads > 0x7fffffffLL * 256 || shards * padded > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)((threads + 255) / 256));
    hipLaunchKernelGGL(unpack_flags_kernel<2>, grid, dim3(256), 0, stream, (const uint8_t *)nullptr, rows, row_len, row_bytes, flags, padded, rows / shards,
                       rows % shards, from);
    return hipGetLastError();
}

hipError_t launch_deinterleave(const float *in, int64_t n_frames, int total, int first, int C, float *out,
                               int64_t out_stride, hipStream_t stream)
{
    if (n_frames <= 0 || C <= 0) return hipSuccess;
    dim3 grid((unsigned)((n_frames + kTileFrames - 1) / kTileFrames), (unsigned)((C + kTileCh - 1) / kTileCh));
    hipLaunchKernelGGL(deinterleave_kernel, grid, dim3(256), 0, stream, in, n_frames, total, first, C, out, out_stride);
    return hipGetLastError();
}

}  // namespace sd
