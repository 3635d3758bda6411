// syldet_sharded.cpp -- one detector bank over several GPUs in ONE process, behind the C ABI.
//
// The reference is one process that owns every channel (Processor.swift:57-59,82,128-141; main.swift:86-89,126-130).  This
// file keeps that shape on a host with several MI355X: a sub-bank (an ordinary syldet_t) and a stream per listed device,
// channels split into contiguous blocks (time-axis ranges when there are fewer channels than devices), every shard's work
// queued before any is waited for, and ONE exchange per batch: the all-gather of the bit-packed detection flags, on RCCL
// communicators made here with ncclCommInitAll.  It is written over the public ABI (syldet_create, syldet_run_device,
// syldet_run, syldet_pack_flags_device) plus the HIP runtime; librccl is loaded when the first bank asks for it.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: the library itself is dlopen'ed (573 MB; one-GPU hosts never load it)

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "kernels.hpp"
#include "syldet_internal.hpp"

using namespace sd;

#define SYLDET_HIP(expr)                                                                         \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(SYLDET_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));  \
    } while (0)

namespace {

// ---- librccl, on demand ------------------------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    std::string why;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy a host process already holds (a torch process carries one under the same soname) is the one we get
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
            r.why = dlerror() ? dlerror() : "dlopen failed";
        }
        if (!r.lib) return;
        auto sym = [&](const char *n) {
            void *p = dlsym(r.lib, n);
            if (!p) r.why = std::string("librccl lacks ") + n;
            return p;
        };
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        if (!r.GetErrorString || !r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd) {
            dlclose(r.lib);
            r.lib = nullptr;
        }
    });
    return &r;
}

#define SYLDET_NCCL(expr)                                                                                      \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess) return fail(SYLDET_ERR_DEVICE, std::string(#expr) + ": " + rccl()->GetErrorString(_r)); \
    } while (0)

struct DevMem {
    void *ptr = nullptr;
    size_t cap = 0;
    int device = 0;
    int reserve(size_t bytes)                                     // (the caller has made `device` current)
    {
        if (bytes <= cap) return SYLDET_OK;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&ptr, bytes);
        if (e != hipSuccess) {
            ptr = nullptr;
            return fail(e == hipErrorOutOfMemory ? SYLDET_ERR_OUT_OF_MEMORY : SYLDET_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
        }
        cap = bytes;
        return SYLDET_OK;
    }
    void release()
    {
        if (ptr) {
            (void)hipSetDevice(device);
            (void)hipFree(ptr);
        }
        ptr = nullptr;
        cap = 0;
    }
};

// dist.shard_channels / dist.shard_plane (the table the process-per-GPU path uses): contiguous channel blocks, the first
// C % G shards one channel longer; with C < G every channel is shared by G / C shards (the first G % C channels by one more).
int shard_table(int32_t C, int32_t G, syldet_shard_t *out)
{
    if (C < 1 || G < 1 || !out) return fail(SYLDET_ERR_INVALID_ARGUMENT, "n_channels and n_shards must be positive");
    if (C >= G) {
        const int32_t base = C / G, extra = C % G;
        for (int32_t r = 0; r < G; r++) out[r] = syldet_shard_t{r, r * base + std::min(r, extra), base + (r < extra ? 1 : 0), 0, 1};
        return SYLDET_OK;
    }
    const int32_t base = G / C, extra = G % C;
    int32_t r = 0;
    for (int32_t ch = 0; ch < C; ch++) {
        const int32_t parts = base + (ch < extra ? 1 : 0);
        for (int32_t part = 0; part < parts; part++, r++) out[r] = syldet_shard_t{r, ch, 1, part, parts};
    }
    return SYLDET_OK;
}

// the calling thread's current device, put back when a call that visits several devices returns
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void shard_evals(int64_t E, int32_t parts, int32_t part, int64_t *first, int64_t *count)
{
    const int64_t base = E / parts, extra = E % parts;
    *first = part * base + std::min<int64_t>(part, extra);
    *count = base + (part < extra ? 1 : 0);
}

}  // namespace

struct syldet_sharded {
    struct Shard {
        syldet_shard_t info{};
        syldet_t *bank = nullptr;
        hipStream_t stream = nullptr;         // the shard's kernels, and nothing else
        hipStream_t xstream = nullptr;        // packing, exchange, unpacking: batch i + 1's kernels do not wait for batch i's collective
        // two sets of exchange buffers, taken in turn (set k of batch i is free again when batch i - 2's exchange has left it)
        hipEvent_t computed[2] = {nullptr, nullptr};   // (compute stream) this shard's kernel of the batch that took set k has finished
        hipEvent_t packed[2] = {nullptr, nullptr};     // (exchange stream) this shard's rows are in send[k]
        hipEvent_t unpacked[2] = {nullptr, nullptr};   // (exchange stream) this device is done with set k: its recv[k], and -- copy exchange -- every shard's send[k]
        DevMem flags[2], send[2], recv[2];    // own flags when the caller keeps none (one per set) | packed rows | every shard's packed rows
        const uint8_t *set_fl[2] = {nullptr, nullptr};   // the flags set k's last batch was packed from (a caller may hand one tensor to every batch)
        ncclComm_t comm = nullptr;
    };
    // One batch call, as the shards' launcher threads see it (the caller's arguments plus what is common to every shard)
    struct Batch {
        const float *const *d_samples = nullptr;
        const int64_t *strides = nullptr;
        float *const *d_outputs = nullptr;
        uint8_t *const *d_flags = nullptr, *const *d_flags_all = nullptr;
        int64_t n_samples = 0, E = 0, chunk = 0;
        int32_t padded_rows = 0;
        int k = 0;
        bool gather = false;
        bool set_is_free = false;             // (copy exchange) every device's last use of buffer set k has finished: no waits for their pulls
        std::vector<uint8_t *> fls;           // per shard: the flags its kernel writes (the shard's own launcher writes the entry)
        std::vector<int64_t> cnts;            // per shard: evaluations of this batch
        // the host-pointer call's arguments (phase HOST)
        const float *samples = nullptr;
        int64_t channel_stride = 0;
        float *outputs = nullptr;
        uint8_t *flags = nullptr;
    };
    // A launcher thread per shard (round 6), alive as long as the bank: its device is current there once and for all, and a
    // batch's per-shard queueing -- the kernel, the packing, the waits and records of the exchange, the unpacking -- runs on all of
    // them at once instead of shard after shard on the caller's thread (0.40 ms a batch for eight shards, against a 0.87 ms kernel).
    // A phase is posted by advancing `ticket`; a launcher spins on it for a while (batches of a stream arrive back to back),
    // then sleeps on the condition variable.
    struct Crew {
        std::vector<std::thread> threads;
        std::mutex m;
        std::condition_variable cv, done_cv;      // a phase is posted | the last launcher has finished it
        std::atomic<uint64_t> ticket{0};
        std::atomic<int> pending{0};
        int phase = 0;
        const Batch *batch = nullptr;
        std::vector<int> status;
        std::vector<std::string> message;
    };
    // (copy exchange) one stream on shard 0's device joins every shard's `packed` event into ONE event per buffer set: a device
    // then waits for one event instead of one per other shard -- 2n waits a batch instead of n (n - 1)
    hipStream_t join = nullptr;
    hipEvent_t all_packed[2] = {nullptr, nullptr};
    std::vector<Shard> shards;
    std::unique_ptr<Crew> crew;               // absent: one shard, SYLDET_SHARDED_INLINE=1, or threads could not be made
    Batch batch;
    syldet_geometry_t geom{};
    int32_t channels = 0, time_range = 0, window = 0;
    int exchange = SYLDET_EXCHANGE_RCCL;
    bool time_mode = false;                   // fewer channels than shards: ranges of evaluations, raw bytes in the exchange
    bool comms_up = false;
    int turn = 0;                             // which set of exchange buffers the next gathering batch takes
    std::mutex mu;                            // one batch call at a time
};

namespace {

void ranges_of(const syldet_sharded *b, const syldet_sharded::Shard &s, int64_t S, int64_t *s0, int64_t *s1, int64_t *e0, int64_t *count)
{
    const int64_t E = syldet_count_evals(s.bank, S);
    int64_t f = 0, n = E > 0 ? E : 0, a = 0, z = S;
    if (s.info.parts > 1) {
        shard_evals(n, s.info.parts, s.info.part, &f, &n);
        // dist.time_shard_samples: from the first frame's hop to the end of the last evaluation's last frame
        a = f * b->geom.hop;
        z = n > 0 ? (f + n + b->time_range - 2) * b->geom.hop + b->geom.gap + b->window : a;
    }
    if (s0) *s0 = a;
    if (s1) *s1 = z;
    if (e0) *e0 = f;
    if (count) *count = n;
}

int bring_up_comms(syldet_sharded *b)
{
    if (b->comms_up || b->exchange != SYLDET_EXCHANGE_RCCL) return SYLDET_OK;
    // (test hook: a host whose RCCL does not come up, on a box where it does -- the callers' fallback to the copy exchange)
    if (const char *e = std::getenv("SYLDET_RCCL_FAIL"))
        if (*e && *e != '0') return fail(SYLDET_ERR_DEVICE, "ncclCommInitAll: refused (SYLDET_RCCL_FAIL is set); create the bank with SYLDET_EXCHANGE_PEER_COPY to do without RCCL");
    Rccl *r = rccl();
    if (!r->lib) return fail(SYLDET_ERR_DEVICE, "librccl could not be loaded (" + r->why + "); create the bank with SYLDET_EXCHANGE_PEER_COPY to do without it");
    const int n = (int)b->shards.size();
    std::vector<int> devs((size_t)n);
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    for (int i = 0; i < n; i++) devs[(size_t)i] = b->shards[(size_t)i].info.device;
    SYLDET_NCCL(r->CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; i++) b->shards[(size_t)i].comm = comms[(size_t)i];
    b->comms_up = true;
    return SYLDET_OK;
}

enum { PHASE_QUEUE = 1, PHASE_FINISH = 2, PHASE_HOST = 3, PHASE_QUIT = -1 };

// Shard i's share of a batch, first half (its device is current): the kernel on the compute stream -- and nothing else there:
// the packing of the flags belongs to the exchange (a kernel of its own between two batches' kernels cost the compute stream
// ~19 us a batch against ~3 us between back-to-back kernels) -- then, on the exchange stream, the packing of its rows.
int phase_queue(syldet_sharded *b, int i)
{
    syldet_sharded::Batch &q = b->batch;
    syldet_sharded::Shard &s = b->shards[(size_t)i];
    const int k = q.k, n = (int)b->shards.size();
    int64_t s0, s1, e0, cnt;
    ranges_of(b, s, q.n_samples, &s0, &s1, &e0, &cnt);
    q.cnts[(size_t)i] = cnt;
    uint8_t *fl = q.d_flags ? q.d_flags[i] : nullptr;
    if (q.gather && !fl) {
        if (int st = s.flags[k].reserve((size_t)s.info.channels * (size_t)std::max<int64_t>(cnt, 1))) return st;
        fl = (uint8_t *)s.flags[k].ptr;
    }
    q.fls[(size_t)i] = fl;
    if (q.gather) {
        if (int st = s.send[k].reserve((size_t)q.chunk)) return st;
        if (int st = s.recv[k].reserve((size_t)q.chunk * (size_t)n)) return st;
    }
    if (cnt > 0) {
        if (!q.d_samples[i]) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL samples for shard " + std::to_string(i));
        // the flags this kernel writes may be the tensor the batch before is still being packed from (a caller that hands the same
        // one to every batch; the library's own are one per set): only then does the kernel wait for that packing
        for (int kk = 0; kk < 2; kk++)
            if (fl && fl == s.set_fl[kk] && hipEventQuery(s.packed[kk]) == hipErrorNotReady)     // (a wait is a packet on the stream: only if needed)
                SYLDET_HIP(hipStreamWaitEvent(s.stream, s.packed[kk], 0));
        if (int st = syldet_run_device(s.bank, q.d_samples[i], s1 - s0, q.strides[i], q.d_outputs ? q.d_outputs[i] : nullptr, fl, s.stream)) return st;
        if (q.gather) {
            SYLDET_HIP(hipEventRecord(s.computed[k], s.stream));
            s.set_fl[k] = fl;
        }
    }
    if (!q.gather) return SYLDET_OK;
    // the exchange, on the exchange stream: the compute stream is free for the next batch's kernel at once
    // (dist.PipelinedFlagGather does the same for the process-per-GPU launcher)
    if (cnt > 0) SYLDET_HIP(hipStreamWaitEvent(s.xstream, s.computed[k], 0));
    // set k's last use was two gathering batches ago.  This shard's own collective and unpacking of that batch are earlier
    // work of this very stream; under the copy exchange the OTHER devices read this send buffer (copies, or their unpacking
    // kernels in place): they must be done with the set first
    // (the caller's thread has asked every device's `unpacked` event of that batch: in a stream of batches it is long complete, and
    // no wait is queued at all)
    if (b->exchange == SYLDET_EXCHANGE_PEER_COPY && !q.set_is_free)
        for (int j = 0; j < n; j++)
            if (j != i) SYLDET_HIP(hipStreamWaitEvent(s.xstream, b->shards[(size_t)j].unpacked[k], 0));
    if (cnt > 0) {
        if (b->time_mode)
            SYLDET_HIP(hipMemcpyAsync(s.send[k].ptr, fl, (size_t)cnt, hipMemcpyDeviceToDevice, s.xstream));
        else
            SYLDET_HIP(launch_pack_flags(fl, s.info.channels, q.E, (uint8_t *)s.send[k].ptr, s.xstream));
    }
    SYLDET_HIP(hipEventRecord(s.packed[k], s.xstream));
    return SYLDET_OK;
}

// Second half, once EVERY shard's first half has been queued (their `packed` events exist; under RCCL the caller's thread has
// issued the grouped all-gather in between): device j pulls every shard's rows (copy exchange), then the gathered rows become
// [C][E] flags, still on its exchange stream.
int phase_finish(syldet_sharded *b, int j)
{
    const syldet_sharded::Batch &q = b->batch;
    syldet_sharded::Shard &d = b->shards[(size_t)j];
    const int k = q.k, n = (int)b->shards.size();
    if (b->exchange != SYLDET_EXCHANGE_RCCL) {
        SYLDET_HIP(hipStreamWaitEvent(d.xstream, b->all_packed[k], 0));      // every shard's rows are packed (joined on the caller's thread)
        // A shard's rows are copied only when they lie on another device; rows on this device (the shard's own; every shard's in a
        // one-GPU rehearsal) are unpacked from where they are.  (Raw stretches of a time-sharded bank, and banks of more shards
        // than the unpacking kernel takes pointers, go through the receive buffer whole.)
        const bool direct = !b->time_mode && n <= kMaxFlagSources;
        FlagSources from;
        for (int i = 0; i < n; i++) {
            syldet_sharded::Shard &s = b->shards[(size_t)i];
            char *dst = (char *)d.recv[k].ptr + (size_t)i * (size_t)q.chunk;
            if (s.info.device != d.info.device)
                SYLDET_HIP(hipMemcpyPeerAsync(dst, d.info.device, s.send[k].ptr, s.info.device, (size_t)q.chunk, d.xstream));
            else if (!direct)
                SYLDET_HIP(hipMemcpyAsync(dst, s.send[k].ptr, (size_t)q.chunk, hipMemcpyDeviceToDevice, d.xstream));
            if (direct) from.p[i] = (const uint8_t *)(s.info.device != d.info.device ? dst : (char *)s.send[k].ptr);
        }
        if (direct) {
            SYLDET_HIP(launch_unpack_flags_from(from, b->channels, q.E, n, q.padded_rows, q.d_flags_all[j], d.xstream));
            SYLDET_HIP(hipEventRecord(d.unpacked[k], d.xstream));
            return SYLDET_OK;
        }
    }
    if (!b->time_mode) {
        SYLDET_HIP(launch_unpack_flags_gathered((const uint8_t *)d.recv[k].ptr, b->channels, q.E, n, q.padded_rows, q.d_flags_all[j], d.xstream));
    } else {
        for (int i = 0; i < n; i++) {
            int64_t e0, cnt;
            ranges_of(b, b->shards[(size_t)i], q.n_samples, nullptr, nullptr, &e0, &cnt);
            if (cnt > 0)
                SYLDET_HIP(hipMemcpyAsync(q.d_flags_all[j] + (size_t)b->shards[(size_t)i].info.first_channel * (size_t)q.E + (size_t)e0,
                                          (const char *)d.recv[k].ptr + (size_t)i * (size_t)q.chunk, (size_t)cnt, hipMemcpyDeviceToDevice, d.xstream));
        }
    }
    SYLDET_HIP(hipEventRecord(d.unpacked[k], d.xstream));
    return SYLDET_OK;
}

// Host buffers: shard i drives its device's pipelined syldet_run on its rows of the caller's arrays (a time-sharded shard: its
// stretch of its channel's row, as a recording of its own).
int phase_host(syldet_sharded *b, int i)
{
    const syldet_sharded::Batch &q = b->batch;
    const syldet_sharded::Shard &s = b->shards[(size_t)i];
    int64_t s0, s1, e0, cnt;
    ranges_of(b, s, q.n_samples, &s0, &s1, &e0, &cnt);
    if (cnt <= 0) return SYLDET_OK;
    const size_t row = (size_t)s.info.first_channel, n_out = (size_t)b->geom.outputs;
    return syldet_run(s.bank, q.samples + row * (size_t)q.channel_stride + s0, s1 - s0, q.channel_stride,
                      q.outputs ? q.outputs + (row * (size_t)q.E + (size_t)e0) * n_out : nullptr,
                      q.flags ? q.flags + row * (size_t)q.E + (size_t)e0 : nullptr);
}

int run_phase_of(syldet_sharded *b, int phase, int i)
{
    return phase == PHASE_QUEUE ? phase_queue(b, i) : phase == PHASE_FINISH ? phase_finish(b, i) : phase_host(b, i);
}

void launcher_main(syldet_sharded *b, int i)
{
    syldet_sharded::Crew &c = *b->crew;
    (void)hipSetDevice(b->shards[(size_t)i].info.device);         // once: the current device is a property of the thread
    uint64_t seen = 0;
    for (;;) {
        // batches of a stream arrive a fraction of a millisecond apart: look for the next one for a while before sleeping
        const auto t0 = std::chrono::steady_clock::now();
        int spins = 0;
        while (c.ticket.load(std::memory_order_acquire) == seen) {
            if ((++spins & 63) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) {
                std::unique_lock<std::mutex> lk(c.m);
                c.cv.wait(lk, [&] { return c.ticket.load(std::memory_order_acquire) != seen; });
                break;
            }
            __builtin_ia32_pause();
        }
        seen = c.ticket.load(std::memory_order_acquire);
        const int phase = c.phase;
        if (phase == PHASE_QUIT) return;
        const int st = run_phase_of(b, phase, i);
        if (st) {
            c.status[(size_t)i] = st;
            c.message[(size_t)i] = syldet_last_error();            // (the error text is per thread)
        }
        if (c.pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {   // the last one out: the caller may be asleep (a long phase)
            std::lock_guard<std::mutex> lk(c.m);
            c.done_cv.notify_one();
        }
    }
}

// One phase of a batch on every shard: on the launcher threads when the bank has them, else shard after shard here.
int run_phase(syldet_sharded *b, int phase)
{
    const int n = (int)b->shards.size();
    if (!b->crew) {
        for (int i = 0; i < n; i++) {
            SYLDET_HIP(hipSetDevice(b->shards[(size_t)i].info.device));
            if (int st = run_phase_of(b, phase, i)) return fail(st, "shard " + std::to_string(i) + ": " + syldet_last_error());
        }
        return SYLDET_OK;
    }
    syldet_sharded::Crew &c = *b->crew;
    std::fill(c.status.begin(), c.status.end(), SYLDET_OK);
    c.phase = phase;
    c.pending.store(n, std::memory_order_relaxed);
    {
        std::lock_guard<std::mutex> lk(c.m);
        c.ticket.fetch_add(1, std::memory_order_release);
    }
    c.cv.notify_all();
    // a device batch's phases take tens of microseconds: the caller looks for their end for a while; the host-pointer call's
    // phase takes as long as the recording: then it sleeps until the last launcher wakes it
    int spins = 0;
    while (c.pending.load(std::memory_order_acquire) != 0) {
        if (++spins > 20000) {
            std::unique_lock<std::mutex> lk(c.m);
            c.done_cv.wait(lk, [&] { return c.pending.load(std::memory_order_acquire) == 0; });
            break;
        }
        __builtin_ia32_pause();
    }
    for (int i = 0; i < n; i++)
        if (c.status[(size_t)i]) return fail(c.status[(size_t)i], "shard " + std::to_string(i) + ": " + c.message[(size_t)i]);
    return SYLDET_OK;
}

void start_crew(syldet_sharded *b)
{
    const int n = (int)b->shards.size();
    const char *e = std::getenv("SYLDET_SHARDED_INLINE");          // (read once, here: A/B runs and the test that holds the two forms together)
    if (n < 2 || (e && *e && *e != '0')) return;
    try {
        std::unique_ptr<syldet_sharded::Crew> c(new syldet_sharded::Crew());
        c->status.assign((size_t)n, SYLDET_OK);
        c->message.resize((size_t)n);
        b->crew = std::move(c);
        for (int i = 0; i < n; i++) b->crew->threads.emplace_back(launcher_main, b, i);
    } catch (...) {
        // fewer threads than shards: none (the ones that exist are sent home), and the caller's thread queues every shard
        if (b->crew) {
            syldet_sharded::Crew &c = *b->crew;
            c.phase = PHASE_QUIT;
            {
                std::lock_guard<std::mutex> lk(c.m);
                c.ticket.fetch_add(1, std::memory_order_release);
            }
            c.cv.notify_all();
            for (auto &t : c.threads) t.join();
            b->crew.reset();
        }
    }
}

void stop_crew(syldet_sharded *b)
{
    if (!b->crew) return;
    syldet_sharded::Crew &c = *b->crew;
    c.phase = PHASE_QUIT;
    {
        std::lock_guard<std::mutex> lk(c.m);
        c.ticket.fetch_add(1, std::memory_order_release);
    }
    c.cv.notify_all();
    for (auto &t : c.threads) t.join();
    b->crew.reset();
}

}  // namespace

extern "C" {

int syldet_shard_table(int32_t n_channels, int32_t n_shards, syldet_shard_t *out) { return shard_table(n_channels, n_shards, out); }

int syldet_shard_evaluations(int64_t n_evals, int32_t parts, int32_t part, int64_t *first, int64_t *count)
{
    if (n_evals < 0 || parts < 1 || part < 0 || part >= parts || !first || !count) return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad sharding arguments");
    shard_evals(n_evals, parts, part, first, count);
    return SYLDET_OK;
}

int syldet_shard_samples(const syldet_config_t *cfg, int64_t first_eval, int64_t count, int64_t *s0, int64_t *s1)
{
    if (!cfg || !s0 || !s1 || first_eval < 0 || count < 0) return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    syldet_geometry_t g;
    if (int st = compute_geometry(*cfg, &g)) return st;
    *s0 = first_eval * g.hop;
    *s1 = count > 0 ? (first_eval + count + cfg->time_range - 2) * g.hop + g.gap + cfg->window_length : *s0;
    return SYLDET_OK;
}

int syldet_create_sharded(const syldet_config_t *cfg, int32_t n_channels, const int32_t *devices, int32_t n_devices, int32_t engine,
                          int32_t exchange, syldet_sharded_t **out)
{
    if (!cfg || !devices || !out) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 1024) return fail(SYLDET_ERR_INVALID_ARGUMENT, "n_devices must be in [1, 1024]");
    if (n_channels < 1) return fail(SYLDET_ERR_INVALID_ARGUMENT, "n_channels must be positive");
    if (exchange != SYLDET_EXCHANGE_RCCL && exchange != SYLDET_EXCHANGE_PEER_COPY) return fail(SYLDET_ERR_INVALID_ARGUMENT, "unknown exchange");
    std::unique_ptr<syldet_sharded> b(new (std::nothrow) syldet_sharded());
    if (!b) return fail(SYLDET_ERR_OUT_OF_MEMORY, "out of memory");
    DeviceGuard restore;
    // RCCL refuses a device listed twice; such a bank (the rehearsal of the shard logic on a one-GPU box) exchanges by copies
    std::vector<int32_t> seen(devices, devices + n_devices);
    std::sort(seen.begin(), seen.end());
    if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) exchange = SYLDET_EXCHANGE_PEER_COPY;
    b->exchange = exchange;
    b->channels = n_channels;
    b->time_range = cfg->time_range;
    b->window = cfg->window_length;
    b->time_mode = n_channels < n_devices;
    std::vector<syldet_shard_t> table((size_t)n_devices);
    if (int st = shard_table(n_channels, n_devices, table.data())) return st;
    try {
        b->shards.resize((size_t)n_devices);
    } catch (const std::bad_alloc &) {
        return fail(SYLDET_ERR_OUT_OF_MEMORY, "out of memory");
    }
    int st = SYLDET_OK;
    for (int32_t i = 0; i < n_devices && st == SYLDET_OK; i++) {
        syldet_sharded::Shard &s = b->shards[(size_t)i];
        s.info = table[(size_t)i];
        s.info.device = devices[i];
        s.flags[0].device = s.flags[1].device = s.send[0].device = s.send[1].device = s.recv[0].device = s.recv[1].device = devices[i];
        st = syldet_create(cfg, s.info.channels, devices[i], engine, &s.bank);      // (validates the device, makes it current)
        if (st) break;
        hipError_t e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.xstream, hipStreamNonBlocking);
        for (int k = 0; k < 2 && e == hipSuccess; k++) {
            e = hipEventCreateWithFlags(&s.packed[k], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.computed[k], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.unpacked[k], hipEventDisableTiming);
        }
        if (e != hipSuccess) st = fail(SYLDET_ERR_DEVICE, std::string("stream / event: ") + hipGetErrorString(e));
    }
    if (st == SYLDET_OK && b->exchange == SYLDET_EXCHANGE_PEER_COPY) {
        hipError_t e = hipSetDevice(b->shards[0].info.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->join, hipStreamNonBlocking);
        for (int k = 0; k < 2 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&b->all_packed[k], hipEventDisableTiming);
        if (e != hipSuccess) st = fail(SYLDET_ERR_DEVICE, std::string("stream / event: ") + hipGetErrorString(e));
    }
    if (st == SYLDET_OK) st = syldet_get_geometry(b->shards[0].bank, &b->geom);
    if (st) {
        const std::string msg = syldet_last_error();               // (the teardown below must not lose the message)
        syldet_sharded_destroy(b.release());
        return fail(st, msg);
    }
    try {
        b->batch.fls.assign((size_t)n_devices, nullptr);
        b->batch.cnts.assign((size_t)n_devices, 0);
    } catch (const std::bad_alloc &) {
        syldet_sharded_destroy(b.release());
        return fail(SYLDET_ERR_OUT_OF_MEMORY, "out of memory");
    }
    start_crew(b.get());
    *out = b.release();
    return SYLDET_OK;
}

int syldet_sharded_destroy(syldet_sharded_t *b)
{
    if (!b) return SYLDET_OK;
    DeviceGuard restore;
    stop_crew(b);
    for (auto &s : b->shards) {
        if (s.stream || s.xstream) (void)hipSetDevice(s.info.device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.xstream) (void)hipStreamSynchronize(s.xstream);
    }
    if (b->join) {
        (void)hipSetDevice(b->shards[0].info.device);
        (void)hipStreamSynchronize(b->join);
        (void)hipStreamDestroy(b->join);
        for (int k = 0; k < 2; k++)
            if (b->all_packed[k]) (void)hipEventDestroy(b->all_packed[k]);
    }
    for (auto &s : b->shards)
        if (s.comm) (void)rccl()->CommDestroy(s.comm);
    for (auto &s : b->shards) {
        (void)hipSetDevice(s.info.device);
        for (int k = 0; k < 2; k++) {
            if (s.packed[k]) (void)hipEventDestroy(s.packed[k]);
            if (s.computed[k]) (void)hipEventDestroy(s.computed[k]);
            s.flags[k].release();
            if (s.unpacked[k]) (void)hipEventDestroy(s.unpacked[k]);
            s.send[k].release();
            s.recv[k].release();
        }
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.xstream) (void)hipStreamDestroy(s.xstream);
        if (s.bank) syldet_destroy(s.bank);
    }
    delete b;
    return SYLDET_OK;
}

int32_t syldet_sharded_channels(const syldet_sharded_t *b) { return b ? b->channels : 0; }
int32_t syldet_sharded_shards(const syldet_sharded_t *b) { return b ? (int32_t)b->shards.size() : 0; }
int32_t syldet_sharded_rccl_ranks(const syldet_sharded_t *b) { return (b && b->exchange == SYLDET_EXCHANGE_RCCL) ? (int32_t)b->shards.size() : 0; }
int32_t syldet_sharded_launcher_threads(const syldet_sharded_t *b) { return (b && b->crew) ? (int32_t)b->crew->threads.size() : 0; }

int syldet_sharded_connect(syldet_sharded_t *b)
{
    if (!b) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL handle");
    std::lock_guard<std::mutex> lock(b->mu);
    DeviceGuard restore;
    if (b->exchange == SYLDET_EXCHANGE_RCCL) return bring_up_comms(b);
    // the copy exchange: direct peer access where the devices offer it (hipMemcpyPeerAsync works either way; through the host
    // without).  Best effort: "already enabled" and "not supported" are both fine.
    for (auto &d : b->shards)
        for (auto &s : b->shards) {
            int can = 0;
            if (s.info.device == d.info.device || hipDeviceCanAccessPeer(&can, d.info.device, s.info.device) != hipSuccess || !can) continue;
            if (hipSetDevice(d.info.device) == hipSuccess) (void)hipDeviceEnablePeerAccess(s.info.device, 0);
        }
    (void)hipGetLastError();
    return SYLDET_OK;
}

int syldet_sharded_shard(const syldet_sharded_t *b, int32_t shard, syldet_shard_t *out)
{
    if (!b || !out || shard < 0 || shard >= (int32_t)b->shards.size()) return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    *out = b->shards[(size_t)shard].info;
    return SYLDET_OK;
}

syldet_t *syldet_sharded_bank(syldet_sharded_t *b, int32_t shard)
{
    return (b && shard >= 0 && shard < (int32_t)b->shards.size()) ? b->shards[(size_t)shard].bank : nullptr;
}

void *syldet_sharded_stream(syldet_sharded_t *b, int32_t shard)
{
    return (b && shard >= 0 && shard < (int32_t)b->shards.size()) ? (void *)b->shards[(size_t)shard].stream : nullptr;
}

void *syldet_sharded_exchange_stream(syldet_sharded_t *b, int32_t shard)
{
    return (b && shard >= 0 && shard < (int32_t)b->shards.size()) ? (void *)b->shards[(size_t)shard].xstream : nullptr;
}

int syldet_sharded_ranges(const syldet_sharded_t *b, int32_t shard, int64_t n_samples, int64_t *s0, int64_t *s1, int64_t *e0, int64_t *count)
{
    if (!b || shard < 0 || shard >= (int32_t)b->shards.size() || n_samples < 0) return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    ranges_of(b, b->shards[(size_t)shard], n_samples, s0, s1, e0, count);
    return SYLDET_OK;
}

// Host buffers: one thread per shard drives that device's pipelined syldet_run on the shard's rows of the caller's arrays
// (for a time-sharded shard: its stretch of its channel's row).  The results of a shard are contiguous in the caller's
// [C][E] layout only when it owns whole rows; a time-sharded shard has one row, and its stretch is contiguous too.
int syldet_sharded_run(syldet_sharded_t *b, const float *samples, int64_t n_samples, int64_t channel_stride, float *outputs, uint8_t *flags)
{
    if (!b) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL handle");
    if (n_samples < 0) return fail(SYLDET_ERR_INVALID_ARGUMENT, "n_samples must be >= 0");
    if (!samples && n_samples > 0) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL samples");
    if (b->channels > 1 && channel_stride < n_samples) return fail(SYLDET_ERR_INVALID_ARGUMENT, "channel_stride must be >= n_samples");
    std::lock_guard<std::mutex> lock(b->mu);
    DeviceGuard restore;                                          // (work(0) runs on the caller's thread and makes shard 0's device current)
    const int64_t E = syldet_count_evals(b->shards[0].bank, n_samples);
    if (E <= 0) return SYLDET_OK;
    syldet_sharded::Batch &q = b->batch;
    q.n_samples = n_samples;
    q.E = E;
    q.samples = samples;
    q.channel_stride = channel_stride;
    q.outputs = outputs;
    q.flags = flags;
    if (b->crew) return run_phase(b, PHASE_HOST);
    // no launcher threads (one shard, or SYLDET_SHARDED_INLINE): a thread per further shard for the length of this call
    const int n = (int)b->shards.size();
    std::vector<int> status((size_t)n, SYLDET_OK);
    std::vector<std::string> message((size_t)n);
    auto work = [&](int i) {
        if (hipSetDevice(b->shards[(size_t)i].info.device) != hipSuccess) {
            status[(size_t)i] = SYLDET_ERR_DEVICE;
            message[(size_t)i] = "hipSetDevice";
            return;
        }
        status[(size_t)i] = phase_host(b, i);
        if (status[(size_t)i]) message[(size_t)i] = syldet_last_error();       // (the error text is per thread)
    };
    std::vector<std::thread> th;
    try {
        for (int i = 1; i < n; i++) th.emplace_back(work, i);
    } catch (...) {
        for (int i = (int)th.size() + 1; i < n; i++) work(i);
    }
    work(0);
    for (auto &t : th) t.join();
    for (int i = 0; i < n; i++)
        if (status[(size_t)i]) return fail(status[(size_t)i], "shard " + std::to_string(i) + ": " + message[(size_t)i]);
    return SYLDET_OK;
}

int syldet_sharded_run_device(syldet_sharded_t *b, const float *const *d_samples, int64_t n_samples, const int64_t *strides,
                              float *const *d_outputs, uint8_t *const *d_flags, uint8_t *const *d_flags_all)
{
    if (!b || !d_samples || !strides) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_samples < 0) return fail(SYLDET_ERR_INVALID_ARGUMENT, "n_samples must be >= 0");
    std::lock_guard<std::mutex> lock(b->mu);
    DeviceGuard restore;
    const int n = (int)b->shards.size();
    const int64_t E = syldet_count_evals(b->shards[0].bank, n_samples);
    if (E <= 0) return SYLDET_OK;
    const bool gather = d_flags_all != nullptr;
    if (gather) {
        for (int i = 0; i < n; i++)
            if (!d_flags_all[i] || ((uintptr_t)d_flags_all[i] & 7)) return fail(SYLDET_ERR_INVALID_ARGUMENT, "d_flags_all entries must be 8-byte aligned device pointers");
        if (int st = bring_up_comms(b)) return st;
    }
    // what travels: bit rows of whole channels (padded to the longest shard), or -- time-sharded -- the raw flags of a stretch
    const int64_t row_bytes = (E + 7) / 8;
    int64_t chunk = 0;                                            // bytes every shard contributes
    int32_t padded_rows = 0;
    if (b->time_mode) {
        for (auto &s : b->shards) {
            int64_t cnt;
            ranges_of(b, s, n_samples, nullptr, nullptr, nullptr, &cnt);
            chunk = std::max(chunk, cnt);
        }
    } else {
        for (auto &s : b->shards) padded_rows = std::max(padded_rows, s.info.channels);
        chunk = (int64_t)padded_rows * row_bytes;                 // (exactly: the unpacking addresses block s at s * padded_rows rows)
    }
    if (b->time_mode) chunk = (chunk + 15) / 16 * 16;

    // The exchange buffers of set k: before any of them is replaced by a longer one, nothing may still be reading it (under the
    // copy exchange OTHER devices' streams pull from a shard's send buffer; hipFree waits for the owning device only)
    const int k = b->turn;
    if (gather) {
        bool grow = false;
        for (int i = 0; i < n; i++) {
            syldet_sharded::Shard &s = b->shards[(size_t)i];
            int64_t cnt;
            ranges_of(b, s, n_samples, nullptr, nullptr, nullptr, &cnt);
            grow = grow || (size_t)chunk > s.send[k].cap || (size_t)chunk * (size_t)n > s.recv[k].cap ||
                   (!(d_flags && d_flags[i]) && (size_t)s.info.channels * (size_t)std::max<int64_t>(cnt, 1) > s.flags[k].cap);
        }
        if (grow) {
            for (auto &s : b->shards) {
                SYLDET_HIP(hipSetDevice(s.info.device));
                SYLDET_HIP(hipStreamSynchronize(s.stream));
                SYLDET_HIP(hipStreamSynchronize(s.xstream));
            }
        }
        b->turn ^= 1;
    }

    syldet_sharded::Batch &q = b->batch;
    q.d_samples = d_samples;
    q.strides = strides;
    q.d_outputs = d_outputs;
    q.d_flags = d_flags;
    q.d_flags_all = d_flags_all;
    q.n_samples = n_samples;
    q.E = E;
    q.chunk = chunk;
    q.padded_rows = padded_rows;
    q.k = k;
    q.gather = gather;
    q.set_is_free = false;
    if (gather && b->exchange == SYLDET_EXCHANGE_PEER_COPY) {
        // set k was last used two gathering batches ago: has every device finished with it (its pulls from the others' send buffers
        // and its unpacking)?  An event never recorded answers "complete".
        q.set_is_free = true;
        for (auto &s : b->shards)
            if (hipEventQuery(s.unpacked[k]) != hipSuccess) {
                q.set_is_free = false;
                break;
            }
        (void)hipGetLastError();                                  // (hipErrorNotReady is an answer, not an error to keep)
    }

    // 1. every shard's kernel on its own device and compute stream and the packing of its flags on its exchange stream: on the
    //    shards' launcher threads, all at once (without them: shard after shard from here)
    if (int st = run_phase(b, PHASE_QUEUE)) return st;
    if (!gather) return SYLDET_OK;

    // 2. the one exchange, on the exchange streams: ONE grouped all-gather, issued from this thread for every communicator
    //    (the copy exchange's pulls are part of step 3: every shard's `packed` event exists by now)
    if (b->exchange == SYLDET_EXCHANGE_RCCL) {
        Rccl *r = rccl();
        SYLDET_NCCL(r->GroupStart());
        for (int i = 0; i < n; i++) {
            syldet_sharded::Shard &s = b->shards[(size_t)i];
            ncclResult_t st = r->AllGather(s.send[k].ptr, s.recv[k].ptr, (size_t)chunk, ncclUint8, s.comm, s.xstream);
            if (st != ncclSuccess) {
                (void)r->GroupEnd();
                return fail(SYLDET_ERR_DEVICE, std::string("ncclAllGather: ") + r->GetErrorString(st));
            }
        }
        SYLDET_NCCL(r->GroupEnd());
    } else {
        SYLDET_HIP(hipSetDevice(b->shards[0].info.device));
        for (auto &s : b->shards) SYLDET_HIP(hipStreamWaitEvent(b->join, s.packed[k], 0));
        SYLDET_HIP(hipEventRecord(b->all_packed[k], b->join));
    }

    // 3. on every device, still on its exchange stream: the gathered rows into [C][E] flags
    return run_phase(b, PHASE_FINISH);
}

int syldet_sharded_synchronize(syldet_sharded_t *b)
{
    if (!b) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL handle");
    DeviceGuard restore;
    for (auto &s : b->shards) {
        SYLDET_HIP(hipSetDevice(s.info.device));
        SYLDET_HIP(hipStreamSynchronize(s.stream));
        SYLDET_HIP(hipStreamSynchronize(s.xstream));
    }
    return SYLDET_OK;
}

}  // extern "C"
