// The streaming front end's sample ring (csrc/sample_ring.hpp, the product's own header) under ThreadSanitizer: one producer
// thread appends ragged buffers the way an audio I/O thread does (AudioInterface.swift:67-70 -> Processor.swift:124), one
// consumer thread does what libsyldet's pump does with them -- reads the tail, copies the pending evaluations' samples out,
// advances head and frames_done -- with the device call replaced by a checksum.  The reference's guarantee is
// TPCircularBuffer's (TPCircularBuffer.h:14: one producer, one consumer, no locks); here it must hold by the atomics alone.
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "sample_ring.hpp"

int main()
{
    const int64_t hop = 132, W = 256, T = 10, total = 3000000;
    sd::ChannelStream cs;
    uint64_t cap = 1;
    while (cap < (uint64_t)(sd::kSampleRingBytes / 4 + W + (T + 1) * hop)) cap <<= 1;
    cs.mask = cap - 1;
    auto value = [](uint64_t i) { return (float)(int)((i * 2654435761u) >> 20 & 0xfff); };
    std::atomic<bool> bad{false};
    int64_t refused = 0;

    std::thread producer([&] {
        std::vector<float> buf(4096);
        uint64_t pos = 0;
        unsigned r = 12345;
        while (pos < (uint64_t)total) {
            r = r * 1664525u + 1013904223u;
            int64_t n = 1 + (r >> 8) % 700;
            if (pos + (uint64_t)n > (uint64_t)total) n = (int64_t)((uint64_t)total - pos);
            for (int64_t i = 0; i < n; i++) buf[(size_t)i] = value(pos + (uint64_t)i);
            if (!cs.has_room(n, hop)) { refused++; std::this_thread::yield(); continue; }     // "Insufficient space on buffer."
            if (!cs.ensure_ring()) { bad = true; return; }
            cs.write(buf.data(), n, 1);
            pos += (uint64_t)n;
        }
    });
    std::thread consumer([&] {
        std::vector<float> out;
        uint64_t seen = 0, rounds = 0;
        while (seen < (uint64_t)total && !bad) {
            const uint64_t tail = cs.tail.load(std::memory_order_acquire);
            // every whole frame is extracted now; every evaluation those samples allow is computed
            const int64_t J = (int64_t)tail >= W ? ((int64_t)tail - W) / hop + 1 : 0;
            cs.frames_done.store(J, std::memory_order_release);
            const uint64_t head = cs.head.load(std::memory_order_relaxed);
            const int64_t avail = (int64_t)(tail - head);
            const int64_t Jl = avail >= W ? (avail - W) / hop + 1 : 0, E = Jl >= T ? Jl - T + 1 : 0;
            if (E > 0) {
                const size_t S = (size_t)(W + (E + T - 2) * hop);
                out.resize(S);
                cs.copy_out(head, out.data(), S);
                for (size_t i = 0; i < S; i++)
                    if (out[i] != value(head + i)) { bad = true; break; }
                cs.head.store(head + (uint64_t)(E * hop), std::memory_order_release);
            }
            seen = tail;
            if (E == 0) std::this_thread::yield();
            if ((++rounds & 255) == 0) std::this_thread::sleep_for(std::chrono::milliseconds(2));   // a slow consumer: the producer meets a full ring
        }
    });
    producer.join();
    consumer.join();
    std::printf("%s refused=%lld\n", bad ? "CORRUPT" : "ok", (long long)refused);
    return bad ? 1 : 0;
}
