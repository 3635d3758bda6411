// sample_ring.hpp -- per-channel streaming state of libsyldet's host side (no device code: the thread-sanitizer build of
// tests/sanitize exercises exactly this header).
//
// The reference keeps, per detector, a 409600-byte sample ring drained one frame at a time and a feature ring drained one
// column per evaluation, both TPCircularBuffers: lock-free single-producer / single-consumer byte rings whose only shared word
// is the fill count (TPCircularBuffer.h:14,102-189; OSAtomicAdd32Barrier at :118,:159).  Here a channel keeps ONE ring of raw
// samples "from the first frame of its next evaluation onward" and a queue of evaluations already computed on the device.
// The producer side (append) touches only the ring's tail and reads head / frames_done -- no lock, and one allocation, on a
// channel's first samples: it can sit on an audio I/O thread like TPCircularBufferProduceBytes does (:177-185).
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <vector>

namespace sd {

constexpr int64_t kSampleRingBytes = 409600;   // CircularShortTimeFourierTransform.init(buffer:) default :61

struct ChannelStream {
    std::vector<float> ring;                 // power-of-two capacity `mask + 1`; allocated by the first append (batch-only banks never pay for it)
    uint64_t mask = 0;
    std::atomic<uint64_t> tail{0};           // producer: total samples ever appended
    std::atomic<uint64_t> head{0};           // consumer: first sample of the next evaluation
    std::atomic<int64_t> frames_done{0};     // STFT frames the reference would have extracted so far
    // consumer side only
    std::deque<std::vector<float>> ready;    // evaluated outputs not yet handed out
    std::vector<float> last;                 // lastOutputs
    std::mutex mu;                           // ready / last (consumer vs. readers of last*; never taken by append)

    // TPCircularBufferProduceBytes fails when fewer than n * 4 bytes are free (TPCircularBuffer.h:177-185); the bytes in the
    // reference's ring are the samples no extracted frame has consumed yet.  Producer side.
    bool has_room(int64_t n, int64_t hop) const
    {
        const uint64_t t = tail.load(std::memory_order_relaxed);
        const int64_t unconsumed = (int64_t)t - frames_done.load(std::memory_order_acquire) * hop;
        if ((unconsumed + n) * 4 > kSampleRingBytes) return false;
        // (cannot overrun the un-evaluated samples: the ring is sized for the bound above plus the evaluation carry)
        return (int64_t)(t - head.load(std::memory_order_acquire)) + n <= (int64_t)(mask + 1);
    }

    // The ring comes into being with a channel's first samples (the only allocation the producer side ever makes); the consumer
    // never looks at it before `tail` says there is something in it.
    bool ensure_ring()
    {
        if (!ring.empty()) return true;
        try {
            ring.assign((size_t)(mask + 1), 0.0f);
        } catch (const std::bad_alloc &) {
            return false;
        }
        return true;
    }

    // n samples, `step` floats apart in the source (1: a plain buffer; the channel count: one channel of interleaved frames).
    // Producer side; the release store of the tail publishes the samples.
    void write(const float *data, int64_t n, int64_t step)
    {
        const uint64_t t = tail.load(std::memory_order_relaxed);
        float *r = ring.data();
        for (int64_t i = 0; i < n; i++) r[(size_t)((t + (uint64_t)i) & mask)] = data[i * step];
        tail.store(t + (uint64_t)n, std::memory_order_release);
    }

    // Consumer side: n samples from absolute position `from` (n > 0 only after an append: the ring exists).
    void copy_out(uint64_t from, float *dst, size_t n) const
    {
        const size_t at = (size_t)(from & mask), first = std::min(n, ring.size() - at);
        std::memcpy(dst, ring.data() + at, first * sizeof(float));
        if (n > first) std::memcpy(dst + first, ring.data(), (n - first) * sizeof(float));
    }
};

}  // namespace sd
