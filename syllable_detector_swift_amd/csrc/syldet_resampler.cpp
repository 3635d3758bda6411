// syldet_resampler.cpp -- ResamplerLinear (Common/Resampler.swift:20-76) for a bank of channels on the
// device, and the stand-alone de-interleave entry point.  The resampling state that depends only on
// sizes (`offset`) lives on the host and is advanced with the reference's own fp32 operations; the
// per-channel carry (`last`) lives on the device next to the data.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <memory>
#include <new>
#include <string>

#include "kernels.hpp"
#include "syldet_internal.hpp"

using namespace sd;

#define SYLDET_HIP(expr)                                                                         \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(SYLDET_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));  \
    } while (0)

struct syldet_resampler {
    double rate_in = 0, rate_out = 0;
    int channels = 0, device = 0;
    float step = 1.0f;               // Float(samplingRateIn / samplingRateOut), :33
    float offset = 0.0f;             // :26
    float *d_last = nullptr;         // [channels], :25
    float *d_in = nullptr, *d_out = nullptr;   // staging of the host-pointer entry point
    size_t in_cap = 0, out_cap = 0;
    hipStream_t stream = nullptr;
};

extern "C" {

int syldet_resampler_create(double rate_in, double rate_out, int32_t n_channels, int32_t device, syldet_resampler_t **out)
{
    if (!out) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (!(rate_in > 0.0) || !(rate_out > 0.0)) return fail(SYLDET_ERR_INVALID_ARGUMENT, "sampling rates must be positive");
    if (n_channels <= 0 || n_channels > 65535) return fail(SYLDET_ERR_INVALID_ARGUMENT, "n_channels must be in [1, 65535]");
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(SYLDET_ERR_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count is 0"));
    if (device < 0 || device >= n_dev) return fail(SYLDET_ERR_NO_DEVICE, "device index out of range");
    std::unique_ptr<syldet_resampler> r(new (std::nothrow) syldet_resampler());
    if (!r) return fail(SYLDET_ERR_OUT_OF_MEMORY, "out of memory");
    r->rate_in = rate_in; r->rate_out = rate_out; r->channels = n_channels; r->device = device;
    r->step = (float)(rate_in / rate_out);
    // device resources; on any failure the partially built handle is torn down like a finished one
    auto bring_up = [&]() -> int {
        SYLDET_HIP(hipSetDevice(device));
        SYLDET_HIP(hipMalloc((void **)&r->d_last, (size_t)n_channels * sizeof(float)));
        SYLDET_HIP(hipMemset(r->d_last, 0, (size_t)n_channels * sizeof(float)));
        SYLDET_HIP(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
        return SYLDET_OK;
    };
    if (int st = bring_up()) {
        syldet_resampler_destroy(r.release());
        return st;
    }
    *out = r.release();
    return SYLDET_OK;
}

int syldet_resampler_destroy(syldet_resampler_t *r)
{
    if (!r) return SYLDET_OK;
    (void)hipSetDevice(r->device);
    if (r->stream) { (void)hipStreamSynchronize(r->stream); (void)hipStreamDestroy(r->stream); }
    if (r->d_last) (void)hipFree(r->d_last);
    if (r->d_in) (void)hipFree(r->d_in);
    if (r->d_out) (void)hipFree(r->d_out);
    delete r;
    return SYLDET_OK;
}

int64_t syldet_resampler_count(const syldet_resampler_t *r, int64_t n_in)
{
    if (!r || n_in <= 0) return 0;
    const int64_t n = (int64_t)(((float)n_in - r->offset) / r->step);    // Int((Float(numSamplesIn) - offset) / step), :40
    return n > 0 ? n : 0;
}

int syldet_resample_device(syldet_resampler_t *r, const float *d_in, int64_t n_in, int64_t in_stride, float *d_out,
                           int64_t out_stride, int64_t *n_out, void *hip_stream)
{
    if (!r) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL handle");
    if (n_out) *n_out = 0;
    if (n_in < 0) return fail(SYLDET_ERR_INVALID_ARGUMENT, "n_in must be >= 0");
    const int64_t n = syldet_resampler_count(r, n_in);
    if (n_in == 0 || n <= 0) return SYLDET_OK;        // nothing to emit; the reference would index an empty array here
    if (!d_in || !d_out) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL buffer");
    if ((r->channels > 1 && in_stride < n_in) || (r->channels > 1 && out_stride < n))
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "row strides must cover the rows");
    SYLDET_HIP(hipSetDevice(r->device));
    SYLDET_HIP(launch_resample_linear(d_in, n_in, in_stride, d_out, n, out_stride, r->channels, r->step, r->offset, r->d_last,
                                      (hipStream_t)hip_stream));
    // offset = indices[numSamplesOut - 1] + step - Float(numSamplesIn - 1), :65, with indices[0] = 0 after :54-56
    float last_index = r->offset + (float)(n - 1) * r->step;
    if (r->offset < 0.0f && n == 1) last_index = 0.0f;
    r->offset = last_index + r->step - (float)(n_in - 1);
    if (n_out) *n_out = n;
    return SYLDET_OK;
}

int syldet_resample(syldet_resampler_t *r, const float *in, int64_t n_in, int64_t in_stride, float *out, int64_t out_stride,
                    int64_t *n_out)
{
    if (!r) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL handle");
    if (n_out) *n_out = 0;
    if (n_in < 0) return fail(SYLDET_ERR_INVALID_ARGUMENT, "n_in must be >= 0");
    const int64_t n = syldet_resampler_count(r, n_in);
    if (n_in == 0 || n <= 0) return SYLDET_OK;
    if (!in || !out) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL buffer");
    SYLDET_HIP(hipSetDevice(r->device));
    const size_t C = (size_t)r->channels, ib = C * (size_t)n_in * sizeof(float), ob = C * (size_t)n * sizeof(float);
    if (ib > r->in_cap) {
        if (r->d_in) (void)hipFree(r->d_in);
        r->d_in = nullptr; r->in_cap = 0;
        SYLDET_HIP(hipMalloc((void **)&r->d_in, ib));
        r->in_cap = ib;
    }
    if (ob > r->out_cap) {
        if (r->d_out) (void)hipFree(r->d_out);
        r->d_out = nullptr; r->out_cap = 0;
        SYLDET_HIP(hipMalloc((void **)&r->d_out, ob));
        r->out_cap = ob;
    }
    SYLDET_HIP(hipMemcpy2DAsync(r->d_in, (size_t)n_in * sizeof(float), in, (size_t)(C > 1 ? in_stride : n_in) * sizeof(float),
                                (size_t)n_in * sizeof(float), C, hipMemcpyHostToDevice, r->stream));
    if (int st = syldet_resample_device(r, r->d_in, n_in, n_in, r->d_out, n, n_out, r->stream)) return st;
    SYLDET_HIP(hipMemcpy2DAsync(out, (size_t)(C > 1 ? out_stride : n) * sizeof(float), r->d_out, (size_t)n * sizeof(float),
                                (size_t)n * sizeof(float), C, hipMemcpyDeviceToHost, r->stream));
    SYLDET_HIP(hipStreamSynchronize(r->stream));
    return SYLDET_OK;
}

int64_t syldet_convert_rate_count(int64_t n_in, double rate_in, double rate_out)
{
    if (n_in <= 0 || !(rate_in > 0.0) || !(rate_out > 0.0)) return 0;
    return (int64_t)((double)(n_in - 1) * rate_out / rate_in) + 1;       // positions i * rate_in / rate_out <= n_in - 1
}

int syldet_convert_rate_device(const float *d_in, int64_t n_in, int64_t in_stride, int32_t n_channels, double rate_in,
                               double rate_out, float *d_out, int64_t out_stride, int64_t *n_out, void *hip_stream)
{
    if (n_out) *n_out = 0;
    if (n_in < 0 || n_channels <= 0 || !(rate_in > 0.0) || !(rate_out > 0.0)) return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    const int64_t n = syldet_convert_rate_count(n_in, rate_in, rate_out);
    if (n <= 0) return SYLDET_OK;
    if (!d_in || !d_out) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (n_channels > 1 && (in_stride < n_in || out_stride < n)) return fail(SYLDET_ERR_INVALID_ARGUMENT, "row strides must cover the rows");
    SYLDET_HIP(launch_convert_rate(d_in, n_in, in_stride, d_out, n, out_stride, n_channels, rate_in / rate_out, (hipStream_t)hip_stream));
    if (n_out) *n_out = n;
    return SYLDET_OK;
}

int syldet_deinterleave_device(const float *d_interleaved, int64_t n_frames, int32_t total_channels, int32_t first_channel,
                               int32_t n_channels, float *d_out, int64_t out_stride, void *hip_stream)
{
    if (n_frames < 0 || total_channels <= 0 || first_channel < 0 || n_channels <= 0 || first_channel + n_channels > total_channels)
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "channel selection outside the interleaved layout");
    if (n_frames == 0) return SYLDET_OK;
    if (!d_interleaved || !d_out) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (n_channels > 1 && out_stride < n_frames) return fail(SYLDET_ERR_INVALID_ARGUMENT, "out_stride must be >= n_frames");
    SYLDET_HIP(launch_deinterleave(d_interleaved, n_frames, total_channels, first_channel, n_channels, d_out, out_stride,
                                   (hipStream_t)hip_stream));
    return SYLDET_OK;
}

int syldet_pack_flags_device(const uint8_t *d_flags, int64_t rows, int64_t row_len, uint8_t *d_bits, void *hip_stream)
{
    if (rows < 0 || row_len < 0 || rows > 65535) return fail(SYLDET_ERR_INVALID_ARGUMENT, "rows must be in [0, 65535], row_len >= 0");
    if (rows == 0 || row_len == 0) return SYLDET_OK;
    if (!d_flags || !d_bits) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL buffer");
    SYLDET_HIP(launch_pack_flags(d_flags, rows, row_len, d_bits, (hipStream_t)hip_stream));
    return SYLDET_OK;
}

int syldet_unpack_flags_device(const uint8_t *d_bits, int64_t rows, int64_t row_len, uint8_t *d_flags, void *hip_stream)
{
    if (rows < 0 || row_len < 0 || rows > 65535) return fail(SYLDET_ERR_INVALID_ARGUMENT, "rows must be in [0, 65535], row_len >= 0");
    if (rows == 0 || row_len == 0) return SYLDET_OK;
    if (!d_flags || !d_bits) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL buffer");
    SYLDET_HIP(launch_unpack_flags(d_bits, rows, row_len, d_flags, (hipStream_t)hip_stream));
    return SYLDET_OK;
}

}  // extern "C"
