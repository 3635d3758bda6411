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
        hipEvent_t pulled[2] = {nullptr, nullptr};     // (exchange stream, copy exchange) this device has read every shard's send[k]
        hipEvent_t unpacked[2] = {nullptr, nullptr};   // (exchange stream) send[k] / recv[k] of this shard are done with
        DevMem flags[2], send[2], recv[2];    // own flags when the caller keeps none (one per set) | packed rows | every shard's packed rows
        const uint8_t *set_fl[2] = {nullptr, nullptr};   // the flags set k's last batch was packed from (a caller may hand one tensor to every batch)
        ncclComm_t comm = nullptr;
    };
    std::vector<Shard> shards;
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
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.pulled[k], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.unpacked[k], hipEventDisableTiming);
        }
        if (e != hipSuccess) st = fail(SYLDET_ERR_DEVICE, std::string("stream / event: ") + hipGetErrorString(e));
    }
    if (st == SYLDET_OK) st = syldet_get_geometry(b->shards[0].bank, &b->geom);
    if (st) {
        const std::string msg = syldet_last_error();               // (the teardown below must not lose the message)
        syldet_sharded_destroy(b.release());
        return fail(st, msg);
    }
    *out = b.release();
    return SYLDET_OK;
}

int syldet_sharded_destroy(syldet_sharded_t *b)
{
    if (!b) return SYLDET_OK;
    DeviceGuard restore;
    for (auto &s : b->shards) {
        if (s.stream || s.xstream) (void)hipSetDevice(s.info.device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.xstream) (void)hipStreamSynchronize(s.xstream);
    }
    for (auto &s : b->shards)
        if (s.comm) (void)rccl()->CommDestroy(s.comm);
    for (auto &s : b->shards) {
        (void)hipSetDevice(s.info.device);
        for (int k = 0; k < 2; k++) {
            if (s.packed[k]) (void)hipEventDestroy(s.packed[k]);
            if (s.computed[k]) (void)hipEventDestroy(s.computed[k]);
            s.flags[k].release();
            if (s.pulled[k]) (void)hipEventDestroy(s.pulled[k]);
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
    const int n = (int)b->shards.size(), n_out = b->geom.outputs;
    std::vector<int> status((size_t)n, SYLDET_OK);
    std::vector<std::string> message((size_t)n);
    auto work = [&](int i) {
        const syldet_sharded::Shard &s = b->shards[(size_t)i];
        int64_t s0, s1, e0, cnt;
        ranges_of(b, s, n_samples, &s0, &s1, &e0, &cnt);
        if (cnt <= 0) return;
        const size_t row = (size_t)s.info.first_channel;
        // a shard of whole rows runs [channels][E]; a time-sharded one runs its single row's stretch as a recording of its own
        status[(size_t)i] = syldet_run(s.bank, samples + row * (size_t)channel_stride + s0, s1 - s0, channel_stride,
                                       outputs ? outputs + (row * (size_t)E + (size_t)e0) * (size_t)n_out : nullptr,
                                       flags ? flags + row * (size_t)E + (size_t)e0 : nullptr);
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

    // 1. every shard's kernels, each on its own device and compute stream: all queued before anything else -- and nothing else
    //    is queued there: the packing of the flags belongs to the exchange (a kernel of its own between two batches' kernels cost
    //    the compute stream ~19 us a batch against ~3 us between back-to-back kernels)
    std::vector<uint8_t *> fls((size_t)n, nullptr);
    std::vector<int64_t> cnts((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        syldet_sharded::Shard &s = b->shards[(size_t)i];
        int64_t s0, s1, e0, cnt;
        ranges_of(b, s, n_samples, &s0, &s1, &e0, &cnt);
        cnts[(size_t)i] = cnt;
        SYLDET_HIP(hipSetDevice(s.info.device));
        uint8_t *fl = d_flags ? d_flags[i] : nullptr;
        if (gather && !fl) {
            if (int st = s.flags[k].reserve((size_t)s.info.channels * (size_t)std::max<int64_t>(cnt, 1))) return st;
            fl = (uint8_t *)s.flags[k].ptr;
        }
        fls[(size_t)i] = fl;
        if (gather) {
            if (int st = s.send[k].reserve((size_t)chunk)) return st;
            if (int st = s.recv[k].reserve((size_t)chunk * (size_t)n)) return st;
        }
        if (cnt <= 0) continue;
        if (!d_samples[i]) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL samples for shard " + std::to_string(i));
        // the flags this kernel writes may be the tensor the batch before is still being packed from (a caller that hands the same
        // one to every batch; the library's own are one per set): only then does the kernel wait for that packing
        for (int kk = 0; kk < 2; kk++)
            if (fl && fl == s.set_fl[kk] && hipEventQuery(s.packed[kk]) == hipErrorNotReady)     // (a wait is a packet on the stream: only if needed)
                SYLDET_HIP(hipStreamWaitEvent(s.stream, s.packed[kk], 0));
        if (int st = syldet_run_device(s.bank, d_samples[i], s1 - s0, strides[i], d_outputs ? d_outputs[i] : nullptr, fl, s.stream)) return st;
        if (gather) {
            SYLDET_HIP(hipEventRecord(s.computed[k], s.stream));
            s.set_fl[k] = fl;
        }
    }
    if (!gather) return SYLDET_OK;

    // 2. the one exchange, on the exchange streams: the compute streams are free for the next batch's kernels at once
    //    (dist.PipelinedFlagGather does the same for the process-per-GPU launcher)
    for (int i = 0; i < n; i++) {
        syldet_sharded::Shard &s = b->shards[(size_t)i];
        SYLDET_HIP(hipSetDevice(s.info.device));
        SYLDET_HIP(hipStreamWaitEvent(s.xstream, s.computed[k], 0));
        // set k's last use was two gathering batches ago.  This shard's own collective and unpacking of that batch are earlier
        // work of this very stream; under the copy exchange the OTHER devices pulled from this send buffer: their pulls first
        if (b->exchange == SYLDET_EXCHANGE_PEER_COPY)
            for (int j = 0; j < n; j++)
                if (j != i) SYLDET_HIP(hipStreamWaitEvent(s.xstream, b->shards[(size_t)j].pulled[k], 0));
        if (cnts[(size_t)i] > 0) {
            if (b->time_mode)
                SYLDET_HIP(hipMemcpyAsync(s.send[k].ptr, fls[(size_t)i], (size_t)cnts[(size_t)i], hipMemcpyDeviceToDevice, s.xstream));
            else
                SYLDET_HIP(launch_pack_flags(fls[(size_t)i], s.info.channels, E, (uint8_t *)s.send[k].ptr, s.xstream));
        }
        SYLDET_HIP(hipEventRecord(s.packed[k], s.xstream));
    }
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
        for (int j = 0; j < n; j++) {                             // device j pulls every shard's rows
            syldet_sharded::Shard &d = b->shards[(size_t)j];
            SYLDET_HIP(hipSetDevice(d.info.device));
            for (int i = 0; i < n; i++) {
                syldet_sharded::Shard &s = b->shards[(size_t)i];
                if (i != j) SYLDET_HIP(hipStreamWaitEvent(d.xstream, s.packed[k], 0));
                char *dst = (char *)d.recv[k].ptr + (size_t)i * (size_t)chunk;
                if (s.info.device == d.info.device)
                    SYLDET_HIP(hipMemcpyAsync(dst, s.send[k].ptr, (size_t)chunk, hipMemcpyDeviceToDevice, d.xstream));
                else
                    SYLDET_HIP(hipMemcpyPeerAsync(dst, d.info.device, s.send[k].ptr, s.info.device, (size_t)chunk, d.xstream));
            }
            SYLDET_HIP(hipEventRecord(d.pulled[k], d.xstream));
        }
    }

    // 3. on every device, still on its exchange stream: the gathered rows into [C][E] flags
    for (int j = 0; j < n; j++) {
        syldet_sharded::Shard &d = b->shards[(size_t)j];
        SYLDET_HIP(hipSetDevice(d.info.device));
        if (!b->time_mode) {
            SYLDET_HIP(launch_unpack_flags_gathered((const uint8_t *)d.recv[k].ptr, b->channels, E, n, padded_rows, d_flags_all[j], d.xstream));
        } else {
            for (int i = 0; i < n; i++) {
                int64_t e0, cnt;
                ranges_of(b, b->shards[(size_t)i], n_samples, nullptr, nullptr, &e0, &cnt);
                if (cnt > 0)
                    SYLDET_HIP(hipMemcpyAsync(d_flags_all[j] + (size_t)b->shards[(size_t)i].info.first_channel * (size_t)E + (size_t)e0,
                                              (const char *)d.recv[k].ptr + (size_t)i * (size_t)chunk, (size_t)cnt, hipMemcpyDeviceToDevice, d.xstream));
            }
        }
        SYLDET_HIP(hipEventRecord(d.unpacked[k], d.xstream));
    }
    return SYLDET_OK;
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
