// syldet.hpp -- C++ mirror of the reference's Swift interface for this path, over the C ABI in
// syldet.h.  Header-only; links against libsyldet.so.
//
// The reference is compiled Swift (no Swift toolchain in the build image), so the host side above
// the C ABI is written in C++ with the reference's names, argument meaning and error behaviour:
//   SyllableDetectorConfig(fromTextFile:)  throws ParseError     Common/SyllableDetectorConfig.swift:170-277
//   SyllableDetector(config:)              fatalError on mismatch Common/SyllableDetector.swift:37-74
//   appendAudioData(_:withSamples:)                               :129-132
//   processNewValue() -> Bool                                     :153-217
//   lastOutputs / lastDetected / seenSyllable()                   :26-31, :220-230
//   ResamplerLinear(fromRate:toRate:).resampleVector / resampleArray  Common/Resampler.swift:20-76
// Swift's fatalError becomes syldetxx::FatalError (a std::runtime_error carrying the status);
// ParseError keeps its four kinds.  A detector here is one channel of a bank; `SyllableDetectorBank`
// is the batched form the MI355X engine is built around.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "syldet.h"

namespace syldetxx {

struct FatalError : std::runtime_error {
    int status;
    FatalError(int st, const std::string &msg) : std::runtime_error(msg), status(st) {}
};

// SyllableDetectorConfig.ParseError, SyllableDetectorConfig.swift:50-55
struct ParseError : std::runtime_error {
    enum Kind { unableToOpenPath, missingValue, invalidValue, mismatchedLength } kind;
    ParseError(Kind k, const std::string &msg) : std::runtime_error(msg), kind(k) {}
};

inline void check(int status)
{
    if (status >= 0) return;
    const std::string msg = std::string(syldet_strerror(status)) + ": " + syldet_last_error();
    switch (status) {
    case SYLDET_ERR_PARSE_OPEN: throw ParseError(ParseError::unableToOpenPath, msg);
    case SYLDET_ERR_PARSE_MISSING: throw ParseError(ParseError::missingValue, msg);
    case SYLDET_ERR_PARSE_INVALID: throw ParseError(ParseError::invalidValue, msg);
    case SYLDET_ERR_PARSE_LENGTH: throw ParseError(ParseError::mismatchedLength, msg);
    default: throw FatalError(status, msg);
    }
}

// SyllableDetectorConfig (struct, SyllableDetectorConfig.swift:11-45): same stored fields.
class SyllableDetectorConfig {
public:
    explicit SyllableDetectorConfig(const std::string &fromTextFile) { check(syldet_config_load_text(fromTextFile.c_str(), &cfg_)); }
    ~SyllableDetectorConfig() { syldet_config_free(cfg_); }
    SyllableDetectorConfig(const SyllableDetectorConfig &) = delete;
    SyllableDetectorConfig &operator=(const SyllableDetectorConfig &) = delete;

    double samplingRate() const { return cfg_->sampling_rate; }
    int fourierLength() const { return cfg_->fourier_length; }
    int windowLength() const { return cfg_->window_length; }
    int windowOverlap() const { return cfg_->window_overlap; }
    std::pair<double, double> freqRange() const { return {cfg_->freq_lo, cfg_->freq_hi}; }
    int timeRange() const { return cfg_->time_range; }
    int spectrogramScaling() const { return cfg_->scaling; }
    std::vector<double> thresholds() const { return std::vector<double>(cfg_->thresholds, cfg_->thresholds + cfg_->n_thresholds); }
    int netInputs() const { return cfg_->layers[0].inputs; }
    int netOutputs() const { return cfg_->layers[cfg_->n_layers - 1].outputs; }
    const syldet_config_t *raw() const { return cfg_; }
    syldet_config_t *raw() { return cfg_; }

private:
    syldet_config_t *cfg_ = nullptr;
};

// A bank of independent detectors on one GPU (Processor.swift:57-59 keeps one SyllableDetector
// per channel; here they share one engine so that a batch of channels is one kernel launch).
class SyllableDetectorBank {
public:
    SyllableDetectorBank(const SyllableDetectorConfig &config, int channels, int device = 0, int engine = SYLDET_ENGINE_AUTO)
    {
        check(syldet_create(config.raw(), channels, device, engine, &h_));
        check(syldet_get_geometry(h_, &geometry_));
    }
    ~SyllableDetectorBank() { syldet_destroy(h_); }
    SyllableDetectorBank(const SyllableDetectorBank &) = delete;
    SyllableDetectorBank &operator=(const SyllableDetectorBank &) = delete;

    const syldet_geometry_t &geometry() const { return geometry_; }
    int channels() const { return syldet_channels(h_); }
    int64_t countEvaluations(int64_t samples) const { return syldet_count_evals(h_, samples); }

    // whole recordings, host buffers: samples [channels][n], outputs [channels][E][outputs], flags [channels][E]
    void run(const float *samples, int64_t n, std::vector<float> &outputs, std::vector<uint8_t> &flags)
    {
        const int64_t E = countEvaluations(n);
        outputs.assign((size_t)channels() * (size_t)E * (size_t)geometry_.outputs, 0.0f);
        flags.assign((size_t)channels() * (size_t)E, 0);
        check(syldet_run(h_, samples, n, n, outputs.data(), flags.data()));
    }
    // device buffers, asynchronous on `hipStream`
    void runDevice(const float *d_samples, int64_t n, int64_t stride, float *d_outputs, uint8_t *d_flags, void *hipStream)
    {
        check(syldet_run_device(h_, d_samples, n, stride, d_outputs, d_flags, hipStream));
    }
    // live use: everything every channel has pending in one device round trip (the consumer loop of
    // Processor.swift:128-141 over all detectors); the detectors' processNewValue() then hand the results out
    int64_t processAll()
    {
        int64_t queued = 0;
        check(syldet_process_all(h_, &queued));
        return queued;
    }
    void appendInterleavedData(const float *data, int64_t frames) { check(syldet_append_interleaved(h_, data, frames, channels())); }
    // fromChannel / ofTotalChannels (CircularShortTimeFourierTransform.swift:203-217): the bank on a subset of a wider stream
    void appendInterleavedData(const float *data, int64_t frames, int32_t totalChannels, const int32_t *fromChannels)
    {
        check(syldet_append_interleaved_channels(h_, data, frames, totalChannels, fromChannels));
    }
    // TrackDetector's sample numbering and debounce (TrackDetector.swift:39-43, :65-100)
    std::vector<int64_t> detections(const uint8_t *flags, int64_t nEvals, double debounceSeconds, int channel)
    {
        std::vector<int64_t> idx((size_t)channels() * (size_t)nEvals), counts((size_t)channels());
        check(syldet_detections(h_, flags, nEvals, debounceSeconds, idx.data(), nEvals, counts.data()));
        return std::vector<int64_t>(idx.begin() + (size_t)channel * (size_t)nEvals,
                                    idx.begin() + (size_t)channel * (size_t)nEvals + (size_t)counts[(size_t)channel]);
    }
    syldet_t *raw() { return h_; }

private:
    syldet_t *h_ = nullptr;
    syldet_geometry_t geometry_{};
};

// One bank over several GPUs of this host, ONE process -- the reference's shape: ProcessorBase.init builds one detector per
// channel and one serial queue drains them all (Processor.swift:57-59,82,128-141; main.swift:86-89,126-130).  The library places
// a sub-bank and a stream on every listed device, splits the channels into contiguous blocks (time-axis ranges with a halo
// when there are fewer channels than devices) and gathers the flags with one RCCL all-gather of their bits per batch.
class SyllableDetectorShardedBank {
public:
    SyllableDetectorShardedBank(const SyllableDetectorConfig &config, int channels, const std::vector<int32_t> &devices,
                                int engine = SYLDET_ENGINE_AUTO, int exchange = SYLDET_EXCHANGE_RCCL)
    {
        check(syldet_create_sharded(config.raw(), channels, devices.data(), (int32_t)devices.size(), engine, exchange, &b_));
        try {
            check(syldet_get_geometry(syldet_sharded_bank(b_, 0), &geometry_));
        } catch (...) {                                            // (no destructor runs for a constructor that throws)
            syldet_sharded_destroy(b_);
            b_ = nullptr;
            throw;
        }
    }
    ~SyllableDetectorShardedBank() { syldet_sharded_destroy(b_); }
    SyllableDetectorShardedBank(const SyllableDetectorShardedBank &) = delete;
    SyllableDetectorShardedBank &operator=(const SyllableDetectorShardedBank &) = delete;

    const syldet_geometry_t &geometry() const { return geometry_; }
    int channels() const { return syldet_sharded_channels(b_); }
    int shards() const { return syldet_sharded_shards(b_); }
    int rcclRanks() const { return syldet_sharded_rccl_ranks(b_); }
    int launcherThreads() const { return syldet_sharded_launcher_threads(b_); }
    // brings the exchange up now (RCCL communicators, or peer access for the copy exchange); throws where the first gathering
    // batch would otherwise have -- a caller that wants to fall back makes the bank again with SYLDET_EXCHANGE_PEER_COPY
    void connect() { check(syldet_sharded_connect(b_)); }
    syldet_shard_t shard(int i) const
    {
        syldet_shard_t s;
        check(syldet_sharded_shard(b_, i, &s));
        return s;
    }
    int64_t countEvaluations(int64_t samples) const { return syldet_count_evals(syldet_sharded_bank(b_, 0), samples); }
    // whole recordings, host buffers: samples [channels][n] -> outputs [channels][E][outputs], flags [channels][E];
    // every device's copies and kernels are in flight together
    void run(const float *samples, int64_t n, std::vector<float> &outputs, std::vector<uint8_t> &flags)
    {
        const int64_t E = countEvaluations(n) > 0 ? countEvaluations(n) : 0;
        outputs.assign((size_t)channels() * (size_t)E * (size_t)geometry_.outputs, 0.0f);
        flags.assign((size_t)channels() * (size_t)E, 0);
        check(syldet_sharded_run(b_, samples, n, n, outputs.data(), flags.data()));
    }
    // per-shard device blocks (syldet_sharded_ranges says which samples shard i reads); flagsAll[i]: every channel's flags on
    // device i through the one exchange.  Asynchronous: synchronize() before reading.
    void runDevice(const float *const *d_samples, int64_t n, const int64_t *strides, float *const *d_outputs, uint8_t *const *d_flags,
                   uint8_t *const *d_flagsAll)
    {
        check(syldet_sharded_run_device(b_, d_samples, n, strides, d_outputs, d_flags, d_flagsAll));
    }
    void synchronize() { check(syldet_sharded_synchronize(b_)); }
    syldet_sharded_t *raw() { return b_; }

private:
    syldet_sharded_t *b_ = nullptr;
    syldet_geometry_t geometry_{};
};

// One channel of a bank with the reference's per-detector surface.
class SyllableDetector {
public:
    SyllableDetector(SyllableDetectorBank &bank, int channel) : bank_(bank), channel_(channel) {}

    void appendAudioData(const float *data, int64_t withSamples) { check(syldet_append(bank_.raw(), channel_, data, withSamples)); }
    bool processNewValue() { const int r = syldet_process_new_value(bank_.raw(), channel_); check(r); return r == 1; }
    std::vector<float> lastOutputs() const
    {
        std::vector<float> out((size_t)bank_.geometry().outputs);
        check(syldet_last_outputs(bank_.raw(), channel_, out.data()));
        return out;
    }
    bool lastDetected() const { const int r = syldet_last_detected(bank_.raw(), channel_); check(r); return r == 1; }
    bool seenSyllable() { const int r = syldet_seen_syllable(bank_.raw(), channel_); check(r); return r == 1; }

private:
    SyllableDetectorBank &bank_;
    int channel_;
};

// ResamplerLinear (Resampler.swift:20-76) for `channels` streams fed in lock-step; state carries over between calls.
class ResamplerLinear {
public:
    ResamplerLinear(double fromRate, double toRate, int channels = 1, int device = 0) : channels_(channels)
    {
        check(syldet_resampler_create(fromRate, toRate, channels, device, &r_));
    }
    ~ResamplerLinear() { syldet_resampler_destroy(r_); }
    ResamplerLinear(const ResamplerLinear &) = delete;
    ResamplerLinear &operator=(const ResamplerLinear &) = delete;

    // host rows [channels][n] -> [channels][returned length]   (resampleArray, :71-75)
    std::vector<float> resampleArray(const std::vector<float> &arr)
    {
        const int64_t n = (int64_t)(arr.size() / (size_t)channels_), m = syldet_resampler_count(r_, n);
        std::vector<float> out((size_t)channels_ * (size_t)(m > 0 ? m : 0));
        int64_t got = 0;
        check(syldet_resample(r_, arr.data(), n, n, out.data(), m > 0 ? m : 1, &got));
        return out;
    }
    // device rows, asynchronous on `hipStream`   (resampleVector, :36-69)
    int64_t resampleVector(const float *d_data, int64_t n, int64_t stride, float *d_out, int64_t out_stride, void *hipStream)
    {
        int64_t got = 0;
        check(syldet_resample_device(r_, d_data, n, stride, d_out, out_stride, &got, hipStream));
        return got;
    }
    int64_t countOutput(int64_t n) const { return syldet_resampler_count(r_, n); }

private:
    syldet_resampler_t *r_ = nullptr;
    int channels_;
};

}  // namespace syldetxx
