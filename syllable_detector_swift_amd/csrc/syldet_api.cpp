// syldet_api.cpp -- the device-facing half of the C ABI: handle life-cycle, table upload,
// kernel dispatch (batch), the streaming front-end, host-buffer conveniences.
//
// There is no CPU execution path in this library: every entry point that computes runs
// HIP kernels on a gfx950 device, and create fails without one.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "fused_plan.hpp"
#include "kernels.hpp"
#include "sample_ring.hpp"
#include "syldet_internal.hpp"

using namespace sd;

namespace {

#define SYLDET_HIP(expr)                                                                         \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(SYLDET_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));  \
    } while (0)

struct DeviceBuffer {
    void *ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return SYLDET_OK;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&ptr, bytes);
        if (e != hipSuccess) {
            ptr = nullptr;
            return fail(e == hipErrorOutOfMemory ? SYLDET_ERR_OUT_OF_MEMORY : SYLDET_ERR_DEVICE,
                        std::string("hipMalloc: ") + hipGetErrorString(e));
        }
        cap = bytes;
        return SYLDET_OK;
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

struct PinnedBuffer {
    void *ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return SYLDET_OK;
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc(&ptr, bytes, hipHostMallocDefault);
        if (e != hipSuccess) {
            ptr = nullptr;
            return fail(SYLDET_ERR_OUT_OF_MEMORY, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        }
        cap = bytes;
        return SYLDET_OK;
    }
    void release()
    {
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

}  // namespace

struct syldet {
    OwnedConfig cfg;
    syldet_geometry_t geom{};
    int channels = 0;
    int device = 0;
    int engine = SYLDET_ENGINE_GENERIC;
    int engine_asked = SYLDET_ENGINE_AUTO;      // what syldet_create was asked for (AUTO commits to the fused engine only for what it can hold to 1e-5)

    // Diagnostic switches (A/B runs, and the tests that hold two forms of a kernel against each other): read from the
    // environment ONCE, when the handle is created -- never on the launch path, and a handle keeps the form it was created for.
    struct Switches {
        bool fused_classic = false;   // SYLDET_FUSED_CLASSIC: the 8-wave fused kernel where both fused kernels take the shape
        bool no_fft1k = false;        // SYLDET_NO_FFT1K: 1024-point frames as two launches
        bool wide_no_front = false;   // SYLDET_WIDE_NO_FRONT: the wide engine's inputs through the preparation kernel and the bf16 image in HBM (rounds 1-2)
        bool wide_wg16 = false;       // SYLDET_WIDE_WG16: the wide engine's GEMM as one workgroup of 16 waves a CU (rounds 1-3), not two of 8
        bool wide_stagger = true;     // SYLDET_WIDE_NOSTAGGER: the two-workgroup GEMM WITHOUT waves 4-7 running one epilogue behind waves 0-3 (round 4's order)
        bool wide_dma_builtin = false;   // SYLDET_WIDE_DMA_BUILTIN: (unstaggered) the weight DMA through the compiler's builtin, not the assembly statement
        bool wide_tiles4 = false;     // SYLDET_WIDE_T4: the staggered GEMM as one workgroup a CU with four evaluation tiles a wave
        bool wide_m32 = false;        // SYLDET_WIDE_M32: the staggered wide GEMM on the 32x32x16 MFMA shape (round 6, an A/B form)
        bool wide_tanh_poly = false;  // SYLDET_WIDE_TANH_POLY: the wide GEMM's hidden TanSig / LogSig layer through a clamped odd polynomial (seven terms, packed fp32), no transcendentals
        bool wide_shape32 = false;    // SYLDET_WIDE_SHAPE32: the wide engine's GEMM on the 32x32x16 MFMA shape (rounds 1-2), not 16x16x32
        bool no_bdft = false;         // SYLDET_NO_BDFT: frames of four hops on the FFT kernels, not the block-transform kernel
        bool no_stft_lanes = false;   // SYLDET_NO_STFT_LANES: the LDS Stockham FFT instead of the lane-butterfly one
        bool no_guard = false;        // SYLDET_NO_GUARD: the precision guard off
        bool no_mlpx = false;         // SYLDET_NO_MLPX: the interpretive network kernels under AUTO
        bool fused_nofold = false;    // SYLDET_FUSED_NOFOLD: not the symmetric-fold kernel (the register-resident-basis / 8-wave kernels)
        bool fused_pad128 = false;    // SYLDET_FUSED_PAD128: hop 128 on the fold kernel's padded pieces, not the staggered chunks (round 6)
        bool fused_nofold2 = false;   // SYLDET_FUSED_NOFOLD2: the fold kernel's once-folded form where the twice-folded one takes the shape
        bool fused_stamps = false;    // SYLDET_FUSED_STAMPS: the stamped diagnostic instantiation
        int fused_ko = 0;             // SYLDET_FUSED_KO=<mask>: knock-out instantiation
        long long host_chunk = 0;     // SYLDET_HOST_CHUNK_BYTES=<n>: bytes of input per stage of the host-pointer pipeline (tests force seams with it)
        void read()
        {
            host_chunk = std::getenv("SYLDET_HOST_CHUNK_BYTES") ? std::atoll(std::getenv("SYLDET_HOST_CHUNK_BYTES")) : 0;
            fused_classic = std::getenv("SYLDET_FUSED_CLASSIC") != nullptr;
            no_fft1k = std::getenv("SYLDET_NO_FFT1K") != nullptr;
            no_bdft = std::getenv("SYLDET_NO_BDFT") != nullptr;
            wide_shape32 = std::getenv("SYLDET_WIDE_SHAPE32") != nullptr;
            wide_wg16 = std::getenv("SYLDET_WIDE_WG16") != nullptr;
            wide_stagger = std::getenv("SYLDET_WIDE_NOSTAGGER") == nullptr;
            wide_dma_builtin = std::getenv("SYLDET_WIDE_DMA_BUILTIN") != nullptr;
            wide_tiles4 = std::getenv("SYLDET_WIDE_T4") != nullptr;
            wide_tanh_poly = std::getenv("SYLDET_WIDE_TANH_POLY") != nullptr;
            wide_m32 = std::getenv("SYLDET_WIDE_M32") != nullptr;
            wide_no_front = std::getenv("SYLDET_WIDE_NO_FRONT") != nullptr;
            no_stft_lanes = std::getenv("SYLDET_NO_STFT_LANES") != nullptr;
            no_guard = std::getenv("SYLDET_NO_GUARD") != nullptr;
            no_mlpx = std::getenv("SYLDET_NO_MLPX") != nullptr;
            fused_nofold = std::getenv("SYLDET_FUSED_NOFOLD") != nullptr;
            fused_nofold2 = std::getenv("SYLDET_FUSED_NOFOLD2") != nullptr;
            fused_pad128 = std::getenv("SYLDET_FUSED_PAD128") != nullptr;
            fused_stamps = std::getenv("SYLDET_FUSED_STAMPS") != nullptr;
            fused_ko = std::getenv("SYLDET_FUSED_KO") ? std::atoi(std::getenv("SYLDET_FUSED_KO")) : 0;
        }
    } sw;

    // device tables
    DeviceBuffer d_window, d_tw, d_sw, d_params, d_thr;
    StftDesc stft{};
    NetDesc net{};

    // scratch
    DeviceBuffer d_columns;           // generic engine: [C][J][F]

    // fused engine tables
    FusedPlan fused;
    DeviceBuffer d_stamps;            // diagnostic stamps (SYLDET_FUSED_STAMPS)
    // precision guard of the fused kernels (kernels.hpp, FixItem): work list + what the exact recomputation needs
    DeviceBuffer d_fix;               // 8 counters | items
    DeviceBuffer d_ctab;              // [N] (cos, sin)(2 pi m / N) fp64
    FixDesc fixd{};
    bool has_fix = false;
    DeviceBuffer d_fused;             // one blob: dfrag | wfrag | koff | bias0 | rvec | w1 | b1 | out_params
    DeviceBuffer d_stage_in, d_stage_out, d_stage_flags, d_stage_idx, d_stage_cnt;
    DeviceBuffer d_planar;            // channel-major copy of interleaved input (syldet_run_interleaved*)

    // the fused engine's DFT front half as the STFT of the other engines (W <= 256, F <= 32, hop % 4 == 0)
    FusedPlan dft;
    DeviceBuffer d_dft;
    bool has_dft = false;

    // the generic engine's network stage on the matrix cores, where the configuration is of its class (kernels_mlpx.hip)
    MlpxPlan mlpx;
    DeviceBuffer d_mlpx;              // afrag | bias0 | w1
    // ... with the block-transform front (kernels_bdft.hip) where frames are four hops long
    BdftPlan bdft;
    DeviceBuffer d_bdft;              // basis | cre | afrag

    // wide-network engine (SYLDET_ENGINE_WIDE_BF16)
    WideDesc wide{};
    DeviceBuffer d_wide;              // packed first-layer chunks | b1 | output maps
    DeviceBuffer d_xn;                // [C*E][kWideK] bf16 normalised inputs
    hipStream_t stream = nullptr;     // used by the host-pointer entry points

    std::vector<std::unique_ptr<ChannelStream>> streams;
    std::mutex pump_mu;               // the staging buffers and the stream below belong to one pump at a time
    PinnedBuffer p_stage_in, p_stage_out;

    // the host-pointer batch call as a pipeline along time (syldet_run): two sets of device buffers, a copy-in and a copy-out
    // stream beside the compute stream, one event of each kind per set
    struct Pipe {
        hipStream_t s_in = nullptr, s_out = nullptr;
        DeviceBuffer d_in[2], d_out[2], d_fl[2];
        hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_d2h[2] = {nullptr, nullptr};
        bool up = false;
    } pipe;
    size_t host_chunk_bytes = (size_t)256 << 20;   // SYLDET_HOST_CHUNK_BYTES (read at create): device staging per set

    // optional per-kernel timing (syldet_profile)
    bool profiling = false;
    // a ring of the last prof_depth batch calls: kMaxTimed kernel slots each, two events a slot
    static constexpr int kMaxTimed = 8;
    struct ProfCall { int count = 0; const char *names[kMaxTimed] = {}; bool fix = false; hipStream_t fix_stream = nullptr; };   // fix: the exact path was launched behind the call's kernels (on fix_stream)
    unsigned *prof_items_dev = nullptr;      // ... as the device addresses it
    unsigned *prof_items = nullptr;          // [prof_items_n] page-locked: the exact path's work items of each profiled call (timed_fixup)
    int prof_items_n = 0;
    std::vector<hipEvent_t> events;          // [prof_depth][kMaxTimed][2], created on first use
    std::vector<ProfCall> prof_calls;        // [prof_depth]
    int prof_depth = 1;
    int64_t prof_seq = 0;                    // batch calls profiled so far
    int prof_cur() const { return (int)((prof_seq > 0 ? prof_seq - 1 : 0) % prof_depth); }
    void prof_begin()                        // a batch call starts
    {
        if (!profiling) return;
        if ((int)prof_calls.size() != prof_depth) prof_calls.assign((size_t)prof_depth, ProfCall());
        prof_seq++;
        prof_calls[(size_t)prof_cur()].count = 0;
        prof_calls[(size_t)prof_cur()].fix = false;
        if (prof_items && prof_cur() < prof_items_n) prof_items[2 * prof_cur()] = prof_items[2 * prof_cur() + 1] = 0u;
    }
};

namespace {


int build_tables(syldet *h)
{
    const syldet_config_t &c = h->cfg.view;
    const syldet_geometry_t &g = h->geom;
    const int N = c.fourier_length, W = c.window_length, M = N / 2;
    int logM = 0;
    while ((1 << logM) < M) logM++;

    std::vector<float> win((size_t)W);
    make_window(c.window, W, win.data());
    std::vector<float2> tw((size_t)std::max(1, M / 2)), sw((size_t)M);
    const double two_pi = 6.283185307179586476925286766559;
    for (int t = 0; t < M / 2; t++) {
        const double a = -two_pi * (double)t / (double)M;
        tw[(size_t)t] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    for (int k = 0; k < M; k++) {
        const double a = -two_pi * (double)k / (double)N;
        sw[(size_t)k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    if (int st = h->d_window.reserve(win.size() * sizeof(float))) return st;
    if (int st = h->d_tw.reserve(tw.size() * sizeof(float2))) return st;
    if (int st = h->d_sw.reserve(sw.size() * sizeof(float2))) return st;
    SYLDET_HIP(hipMemcpy(h->d_window.ptr, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice));
    SYLDET_HIP(hipMemcpy(h->d_tw.ptr, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    SYLDET_HIP(hipMemcpy(h->d_sw.ptr, sw.data(), sw.size() * sizeof(float2), hipMemcpyHostToDevice));

    StftDesc &s = h->stft;
    s.N = N; s.W = W; s.M = M; s.logM = logM;
    s.hop = g.hop; s.gap = g.gap; s.f0 = g.f0; s.F = g.bins;
    s.power_mode = c.spectrum == SYLDET_SPECTRUM_MAGNITUDE ? 1 : 0;
    s.window = (const float *)h->d_window.ptr;
    s.tw = (const float2 *)h->d_tw.ptr;
    s.sw = (const float2 *)h->d_sw.ptr;

    // parameter blob for the unfolded network
    NetDesc &n = h->net;
    std::memset(&n, 0, sizeof(n));
    if (c.n_input_fns > kMaxFns || c.n_output_fns > kMaxFns || c.n_layers > kMaxLayers)
        return fail(SYLDET_ERR_UNSUPPORTED, "more than 8 processing functions or 8 layers");
    std::vector<float> blob;
    auto push = [&blob](const float *p, size_t cnt) {
        const int off = (int)blob.size();
        blob.insert(blob.end(), p, p + cnt);
        return off;
    };
    n.n_in_fns = c.n_input_fns;
    for (int i = 0; i < c.n_input_fns; i++) {
        const syldet_fn_t &f = c.input_fns[i];
        n.in_fns[i].kind = f.kind;
        n.in_fns[i].y = f.y;
        if (f.count > 0) {
            n.in_fns[i].xoff = push(f.x_offsets, (size_t)f.count);
            n.in_fns[i].gain = push(f.gains, (size_t)f.count);
        }
    }
    n.n_layers = c.n_layers;
    int max_width = g.inputs;
    for (int l = 0; l < c.n_layers; l++) {
        const syldet_layer_t &L = c.layers[l];
        n.layers[l].in = L.inputs;
        n.layers[l].out = L.outputs;
        n.layers[l].tf = L.transfer;
        n.layers[l].w = push(L.weights, (size_t)L.inputs * (size_t)L.outputs);
        n.layers[l].b = push(L.biases, (size_t)L.outputs);
        max_width = std::max(max_width, L.outputs);
    }
    n.n_out_fns = c.n_output_fns;
    for (int i = 0; i < c.n_output_fns; i++) {
        const syldet_fn_t &f = c.output_fns[i];
        n.out_fns[i].kind = f.kind;
        n.out_fns[i].y = f.y;
        n.out_fns[i].xoff = push(f.x_offsets, (size_t)f.count);
        n.out_fns[i].gain = push(f.gains, (size_t)f.count);
    }
    n.I = g.inputs;
    n.n_out = g.outputs;
    n.max_width = max_width;
    n.scaling = c.scaling;
    n.rule = c.rule;
    if ((size_t)max_width * 2 * 4 * sizeof(float) > 160 * 1024)
        return fail(SYLDET_ERR_UNSUPPORTED, "layer wider than the generic engine's LDS budget");
    if (int st = h->d_params.reserve(std::max<size_t>(blob.size(), 1) * sizeof(float))) return st;
    if (!blob.empty())
        SYLDET_HIP(hipMemcpy(h->d_params.ptr, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    if (int st = h->d_thr.reserve((size_t)c.n_thresholds * sizeof(double))) return st;
    SYLDET_HIP(hipMemcpy(h->d_thr.ptr, c.thresholds, (size_t)c.n_thresholds * sizeof(double), hipMemcpyHostToDevice));
    n.params = (const float *)h->d_params.ptr;
    n.thresholds = (const double *)h->d_thr.ptr;
    return SYLDET_OK;
}

// float -> bf16, round to nearest even (what v_cvt_pk_bf16_f32 does on the device)
uint16_t to_bf16(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// Wide-network engine tables: the first layer in MFMA A-operand order, 32 hidden units per chunk (kernels_wide.hip).
int upload_wide(syldet *h, std::string &why)
{
    const syldet_config_t &c = h->cfg.view;
    if (c.n_layers != 2) { why = "needs exactly two layers"; return SYLDET_ERR_UNSUPPORTED; }
    const syldet_layer_t &L0 = c.layers[0], &L1 = c.layers[1];
    if (L0.inputs > kWideK) { why = "more than 320 network inputs"; return SYLDET_ERR_UNSUPPORTED; }
    if (L0.outputs < 32) { why = "hidden layer narrower than 32 units"; return SYLDET_ERR_UNSUPPORTED; }
    if (L1.outputs > 4) { why = "more than 4 outputs"; return SYLDET_ERR_UNSUPPORTED; }
    const int I = L0.inputs, H = L0.outputs, n_out = L1.outputs, n_chunks = (H + 31) / 32;
    const size_t chunk_u16 = kWideChunkBytes / 2;
    std::vector<uint16_t> pack((size_t)n_chunks * chunk_u16, 0);
    // TanSig / LogSig hidden layers are folded into the tables: tanh(x) = 1 - 2 / (2^(s x) + 1) with s = 2 log2(e),
    // logsig(x) = 1 / (2^(s x) + 1) with s = -log2(e).  The kernel then computes r = 1 / (2^acc + 1) with acc = s (W0 x + b0)
    // and y += w1' r, where w1' = -2 w1 and b1' = b1 + sum w1 for tanh, unchanged for logsig.
    const bool sig = L0.transfer == SYLDET_TF_TANSIG || L0.transfer == SYLDET_TF_LOGSIG;
    const bool shape16 = !h->sw.wide_shape32;
    // The input chain the training script writes -- [l2normalize,] affine maps, on linear columns -- needs no preparation pass:
    // u_i = a_i (v_i r) + o_i with r = 1 / |v| (or 1), so W0 u + b0 = (W0 diag a) (v r) + (W0 o + b0): the affine part goes
    // into the first layer here, and the GEMM kernel makes its operands bf16(v r) from the columns themselves.  (Also the
    // better numbers: bf16 keeps 8 bits of v r instead of 8 bits of a value that sits next to -1.)
    bool front = shape16 && !h->sw.wide_no_front && c.scaling == SYLDET_SCALING_LINEAR && wide_front_fits(h->geom.bins, I);
    int l2 = 0;
    std::vector<double> fa((size_t)I, 1.0), fo((size_t)I, 0.0);
    for (int k = 0; k < c.n_input_fns && front; k++) {
        const syldet_fn_t &f = c.input_fns[k];
        if (f.kind == SYLDET_FN_L2NORMALIZE && k == 0) { l2 = 1; continue; }
        if (f.kind != SYLDET_FN_MAPMINMAX && f.kind != SYLDET_FN_MAPSTD) { front = false; break; }
        for (int i = 0; i < I; i++) {                              // MapMinMax.apply :127-131, MapStd.apply :162-169
            fo[(size_t)i] = (fo[(size_t)i] - (double)f.x_offsets[i]) * (double)f.gains[i] + (double)f.y;
            fa[(size_t)i] *= (double)f.gains[i];
        }
    }
    // The fold quantises v r (not u = a v r + o) to bf16: an error of 2^-9 |a_i v_i r| in u_i.  For maps trained on a range
    // around zero (the example net: offsets of 1e-5, gains of 4) that is the better deal; for a range away from zero
    // (x in [0.05, 0.06]: |a x| = 11 against |u| <= 1) the same relative error is |o_i| times larger in u than bf16(u) would
    // leave, and the 1e-2 bar goes.  |o_i| = |u_i - a_i v_i r| bounds |a_i v_i r| wherever u is O(1): such chains keep the
    // preparation kernel, which rounds u itself.
    if (front) {
        double worst = 0.0;
        for (int i = 0; i < I; i++) worst = std::max(worst, std::fabs(fo[(size_t)i]));
        if (!(worst <= 4.0)) front = false;
    }
    // SYLDET_WIDE_TANH_POLY (round 6, an A/B form): the kernel evaluates t = tanh_poly(acc) -- a clamped odd polynomial, no
    // transcendental -- and y += w1' t.  tanh: acc = W0 x + b0, w1' = w1; logsig(x) = 1/2 + tanh(x / 2) / 2: acc = (W0 x + b0) / 2,
    // w1' = w1 / 2, b1' = b1 + sum w1 / 2.
    // SYLDET_WIDE_M32: the chunks in the 32x32x16 order for wide_gemm32s_kernel (one output, the front end, 256 evaluations' columns
    // and two chunk buffers in half a CU's LDS: launch_wide_gemm checks the same)
    const bool m32 = shape16 && front && n_out == 1 && h->sw.wide_m32 && !h->sw.wide_tanh_poly;
    const bool pack16 = shape16 && !m32;
    const bool poly = sig && shape16 && front && n_out == 1 && h->sw.wide_tanh_poly;   // (the forms that have an instantiation: kernels_wide.hip)
    const double sc = !sig ? 1.0 : poly ? (L0.transfer == SYLDET_TF_TANSIG ? 1.0 : 0.5)
                                        : (L0.transfer == SYLDET_TF_TANSIG ? 2.8853900817779268 : -1.4426950408889634);
    const double w1s = !sig ? 1.0 : poly ? (L0.transfer == SYLDET_TF_TANSIG ? 1.0 : 0.5) : (L0.transfer == SYLDET_TF_TANSIG ? -2.0 : 1.0);
    for (int ch = 0; ch < n_chunks; ch++) {
        uint16_t *frag = pack.data() + (size_t)ch * chunk_u16;
        for (int ks = 0; ks < kWideK / 16; ks++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    // 32x32x16: fragment ks = k-step of 16, lane l = unit l % 32, k = 8 (l / 32) + j
                    // 16x16x32: fragment ks = (k-step of 32, unit tile of 16), lane l = unit l % 16 of its tile, k = 8 (l / 16) + j
                    const int unit = pack16 ? 32 * ch + 16 * (ks & 1) + (l & 15) : 32 * ch + (l & 31);
                    const int k = pack16 ? 32 * (ks >> 1) + 8 * (l >> 4) + j : 16 * ks + 8 * (l >> 5) + j;
                    const float v = (unit < H && k < I) ? (float)(sc * (double)L0.weights[(size_t)unit * I + k] * (front ? fa[(size_t)k] : 1.0)) : 0.0f;
                    frag[((size_t)ks * 64 + l) * 8 + j] = to_bf16(v);
                }
        float *cst = reinterpret_cast<float *>(frag + (size_t)(kWideK / 16) * 64 * 8);
        for (int u = 0; u < 32; u++) {
            const int unit = 32 * ch + u;
            double b0f = unit < H ? (double)L0.biases[unit] : 0.0;
            if (front && unit < H)
                for (int i = 0; i < I; i++) b0f += (double)L0.weights[(size_t)unit * I + i] * fo[(size_t)i];
            cst[u] = unit < H ? (float)(sc * b0f) : 0.0f;
            for (int o = 0; o < 4; o++) cst[32 + 32 * o + u] = (unit < H && o < n_out) ? (float)(w1s * (double)L1.weights[(size_t)o * H + unit]) : 0.0f;
        }
    }
    std::vector<float> misc((size_t)n_out);
    for (int o = 0; o < n_out; o++) {
        double b = (double)L1.biases[o];
        if (sig && !poly && L0.transfer == SYLDET_TF_TANSIG)
            for (int u = 0; u < H; u++) b += (double)L1.weights[(size_t)o * H + u];
        if (poly && L0.transfer == SYLDET_TF_LOGSIG)
            for (int u = 0; u < H; u++) b += 0.5 * (double)L1.weights[(size_t)o * H + u];
        misc[(size_t)o] = (float)b;
    }
    for (int k = 0; k < c.n_output_fns; k++) {
        const syldet_fn_t &f = c.output_fns[k];
        misc.push_back(f.y);
        misc.insert(misc.end(), f.gains, f.gains + n_out);
        misc.insert(misc.end(), f.x_offsets, f.x_offsets + n_out);
    }
    const size_t pack_bytes = pack.size() * 2, misc_bytes = misc.size() * sizeof(float);
    if (int st = h->d_wide.reserve(pack_bytes + misc_bytes)) return st;
    SYLDET_HIP(hipMemcpy(h->d_wide.ptr, pack.data(), pack_bytes, hipMemcpyHostToDevice));
    SYLDET_HIP(hipMemcpy((char *)h->d_wide.ptr + pack_bytes, misc.data(), misc_bytes, hipMemcpyHostToDevice));
    WideDesc &d = h->wide;
    d.H = H; d.n_chunks = n_chunks; d.n_out = n_out; d.tf0 = L0.transfer; d.tf1 = L1.transfer; d.rule = c.rule;
    d.n_out_fns = c.n_output_fns;
    d.sig = sig ? 1 : 0;
    d.poly = poly ? 1 : 0;
    d.m32 = m32 ? 1 : 0;
    d.shape16 = shape16 ? 1 : 0;
    d.front = front ? 1 : 0; d.l2 = l2; d.I = I; d.F = h->geom.bins;
    d.wg8 = h->sw.wide_wg16 ? 0 : 1;
    d.stagger = h->sw.wide_stagger ? 1 : 0;
    d.dma_builtin = h->sw.wide_dma_builtin ? 1 : 0;
    d.tiles4 = h->sw.wide_tiles4 ? 1 : 0;
    d.wpack = (const uint4 *)h->d_wide.ptr;
    d.b1 = (const float *)((const char *)h->d_wide.ptr + pack_bytes);
    d.out_params = d.b1 + n_out;
    d.thresholds = (const double *)h->d_thr.ptr;
    return SYLDET_OK;
}

int upload_plan(syldet *h, FusedPlan &p, DeviceBuffer &buf)
{
    std::vector<unsigned char> blob;
    auto put = [&blob](const void *src, size_t bytes) {
        const size_t off = (blob.size() + 255) / 256 * 256;
        blob.resize(off + std::max<size_t>(bytes, 16));
        if (bytes) std::memcpy(blob.data() + off, src, bytes);
        return off;
    };
    const size_t o_d = put(p.dfrag.data(), p.dfrag.size() * 2), o_w = put(p.afrag.data(), p.afrag.size() * 2);
    const size_t o_k = put(p.koff.data(), p.koff.size() * 4), o_b = put(p.bias0.data(), p.bias0.size() * 4);
    const size_t o_r = put(p.rvec.data(), p.rvec.size() * 4), o_w1 = put(p.w1.data(), p.w1.size() * 4);
    const size_t o_b1 = put(p.b1.data(), p.b1.size() * 4), o_op = put(p.out_params.data(), p.out_params.size() * 4);
    const size_t o_wt = put(p.afrag_t.data(), p.afrag_t.size() * 2);
    const size_t o_sf = put(p.sfrag.data(), p.sfrag.size() * 2), o_sl = put(p.slone.data(), p.slone.size() * 4);
    const size_t o_ww = put(p.afrag_w.data(), p.afrag_w.size() * 2);
    const size_t o_s2 = put(p.sfrag2.data(), p.sfrag2.size() * 2), o_w2 = put(p.swin2.data(), p.swin2.size() * 4);
    const size_t o_c2 = put(p.s2c.data(), p.s2c.size() * 4), o_t2 = put(p.afrag_t2.data(), p.afrag_t2.size() * 2);
    const size_t o_x2 = put(p.afrag_w2.data(), p.afrag_w2.size() * 2);
    if (int st = buf.reserve(blob.size())) return st;
    SYLDET_HIP(hipMemcpy(buf.ptr, blob.data(), blob.size(), hipMemcpyHostToDevice));
    unsigned char *base = (unsigned char *)buf.ptr;
    FusedDesc &d = p.desc;
    d.dfrag = (const uint4 *)(base + o_d);
    d.afrag = (const uint4 *)(base + o_w);
    d.afrag_t = (const uint4 *)(base + o_wt);
    d.sfrag = (const uint4 *)(base + o_sf);
    d.slone = (const float *)(base + o_sl);
    d.afrag_w = (const uint4 *)(base + o_ww);
    d.sfrag2 = (const uint4 *)(base + o_s2);
    d.swin2 = (const float *)(base + o_w2);
    d.s2c = (const float *)(base + o_c2);
    d.afrag_t2 = (const uint4 *)(base + o_t2);
    d.afrag_w2 = (const uint4 *)(base + o_x2);
    d.koff = (const int *)(base + o_k);
    d.bias0 = (const float *)(base + o_b);
    d.rvec = (const float *)(base + o_r);
    d.w1 = (const float *)(base + o_w1);
    d.b1 = (const float *)(base + o_b1);
    d.out_params = (const float *)(base + o_op);
    d.thresholds = (const double *)h->d_thr.ptr;
    return SYLDET_OK;
}

int upload_fused(syldet *h) { return upload_plan(h, h->fused, h->d_fused); }

// What fixup_kernel (the exact recomputation behind the fused kernels' precision guard) needs: geometry, the fp32 window
// table, a trigonometric table in fp64.
int upload_fix(syldet *h)
{
    const syldet_config_t &c = h->cfg.view;
    const int N = c.fourier_length;
    std::vector<double> tab((size_t)N * 2);
    const double two_pi = 6.283185307179586476925286766559;
    for (int m = 0; m < N; m++) {
        tab[(size_t)2 * m] = std::cos(two_pi * (double)m / (double)N);
        tab[(size_t)2 * m + 1] = std::sin(two_pi * (double)m / (double)N);
    }
    if (int st = h->d_ctab.reserve(tab.size() * sizeof(double))) return st;
    SYLDET_HIP(hipMemcpy(h->d_ctab.ptr, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    FixDesc &f = h->fixd;
    f.N = N; f.W = c.window_length; f.hop = h->geom.hop; f.gap = h->geom.gap; f.f0 = h->geom.f0; f.F = h->geom.bins; f.T = c.time_range;
    f.power_mode = c.spectrum == SYLDET_SPECTRUM_MAGNITUDE ? 1 : 0;
    f.window = (const float *)h->d_window.ptr;
    f.ctab = (const double2 *)h->d_ctab.ptr;
    h->has_fix = true;
    return SYLDET_OK;
}

// The work list for one launch over `units` evaluations (or frames) per channel: room for every 16-unit tile of every
// segment of every channel, so the list cannot overflow; the counters are zeroed when the buffer is (re)allocated and by the
// recomputation kernel itself after every launch.  SYLDET_NO_GUARD=1 (diagnostic A/B only) turns the guard off.
int prepare_fix(syldet *h, int C, int64_t units, int64_t segments, hipStream_t stream, FixList &out)
{
    out = FixList{nullptr, nullptr, 0};
    if (!h->has_fix || h->sw.no_guard) return SYLDET_OK;
    // (a list for every tile of every segment cannot overflow; past 2^26 items -- a gigabyte of list, batches of 10^9
    // evaluations -- it is bounded instead, and a kernel that finds it full raises the sticky overflow flag that
    // syldet_fixup_stats reports: the guard is never silently off)
    uint64_t cap = (uint64_t)C * (uint64_t)((units + 15) / 16 + 8 * segments + 16);
    if (cap > (1ull << 26)) cap = 1ull << 26;
    const size_t bytes = 32 + (size_t)cap * sizeof(FixItem);
    if (bytes > h->d_fix.cap) {
        if (int st = h->d_fix.reserve(bytes + bytes / 4)) return st;
        SYLDET_HIP(hipMemsetAsync(h->d_fix.ptr, 0, 32, stream));
    }
    out.counters = (unsigned *)h->d_fix.ptr;
    out.items = (FixItem *)((char *)h->d_fix.ptr + 32);
    out.capacity = (unsigned)cap;
    return SYLDET_OK;
}

int upload_mlpx(syldet *h)
{
    MlpxPlan &p = h->mlpx;
    const size_t a_bytes = (p.afrag.size() * 2 + 255) / 256 * 256;
    std::vector<unsigned char> blob(a_bytes + 64);
    std::memcpy(blob.data(), p.afrag.data(), p.afrag.size() * 2);
    std::memcpy(blob.data() + a_bytes, p.bias0.data(), 16);
    std::memcpy(blob.data() + a_bytes + 16, p.w1.data(), 16);
    if (int st = h->d_mlpx.reserve(blob.size())) return st;
    SYLDET_HIP(hipMemcpy(h->d_mlpx.ptr, blob.data(), blob.size(), hipMemcpyHostToDevice));
    unsigned char *base = (unsigned char *)h->d_mlpx.ptr;
    p.desc.afrag = (const uint4 *)base;
    p.desc.bias0 = (const float *)(base + a_bytes);
    p.desc.w1 = (const float *)(base + a_bytes + 16);
    p.desc.thresholds = (const double *)h->d_thr.ptr;
    return SYLDET_OK;
}

int upload_bdft(syldet *h)
{
    BdftPlan &p = h->bdft;
    auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_bytes = pad(p.basis.size() * 2), c_bytes = pad(p.cre.size() * 4), a_bytes = pad(p.afrag.size() * 2);
    std::vector<unsigned char> blob(b_bytes + c_bytes + a_bytes);
    std::memcpy(blob.data(), p.basis.data(), p.basis.size() * 2);
    std::memcpy(blob.data() + b_bytes, p.cre.data(), p.cre.size() * 4);
    std::memcpy(blob.data() + b_bytes + c_bytes, p.afrag.data(), p.afrag.size() * 2);
    if (int st = h->d_bdft.reserve(blob.size())) return st;
    SYLDET_HIP(hipMemcpy(h->d_bdft.ptr, blob.data(), blob.size(), hipMemcpyHostToDevice));
    unsigned char *base = (unsigned char *)h->d_bdft.ptr;
    p.desc.basis = (const uint4 *)base;
    p.desc.cre = (const float *)(base + b_bytes);
    p.desc.afrag = (const uint4 *)(base + b_bytes + c_bytes);
    p.md = h->mlpx.desc;                 // (its device pointers are in place: upload_mlpx ran first)
    p.md.KB = 4;
    p.md.col_stride = 32 * 4 + 8;
    return SYLDET_OK;
}

// A plan for the DFT front half alone: the real STFT geometry with a one-frame, one-unit stand-in network (the
// spectrogram instantiation never touches the network tables).
int build_dft_plan(syldet *h)
{
    const syldet_config_t &c = h->cfg.view;
    syldet_config_t sc = c;
    const int F = h->geom.bins;
    // (two layers, F -> 1 -> 1: the class the symmetric-fold kernel's plan is made for)
    std::vector<float> w((size_t)F, 0.0f), b(1, 0.0f), w2(1, 1.0f);
    syldet_layer_t layers[2] = {};
    layers[0].inputs = F; layers[0].outputs = 1; layers[0].transfer = SYLDET_TF_PURELIN; layers[0].weights = w.data(); layers[0].biases = b.data();
    layers[1].inputs = 1; layers[1].outputs = 1; layers[1].transfer = SYLDET_TF_PURELIN; layers[1].weights = w2.data(); layers[1].biases = b.data();
    double thr = 0.0;
    sc.time_range = 1; sc.scaling = SYLDET_SCALING_LINEAR; sc.spectrum = SYLDET_SPECTRUM_POWER;
    sc.n_input_fns = 0; sc.input_fns = nullptr; sc.n_output_fns = 0; sc.output_fns = nullptr;
    sc.n_layers = 2; sc.layers = layers; sc.n_thresholds = 1; sc.thresholds = &thr;
    syldet_geometry_t sg = h->geom;
    sg.inputs = F; sg.outputs = 1;
    if (!make_fused_plan(sc, sg, h->dft)) return SYLDET_OK;          // not applicable: the generic FFT stays
    // (two spectrogram instantiations: the fold kernel's twice-folded form for 256-point frames under a 256-sample window,
    // the 8-wave kernel's for its shapes)
    if (!h->dft.desc.classic_ok && !fused_s_spectrogram_applicable(h->dft.desc)) return SYLDET_OK;
    if (int st = upload_plan(h, h->dft, h->d_dft)) return st;
    h->dft.desc.spect_power = c.spectrum == SYLDET_SPECTRUM_MAGNITUDE ? 1 : 0;
    h->has_dft = true;
    return SYLDET_OK;
}

// Does the network's input chain start with a scale-invariant normaliser (l2normalize, normalize, normalizestd:
// NeuralNet.swift:41-109)?  Behind one, the network sees a unit-scale vector whatever the recording's level, and the
// matrix-core transforms' error -- 2^-21.4 of a frame's column level -- is 2^-21.4 of that unit.  Without one the network sees
// the columns at the recording's level, and the same relative error grows with it where an fp32 FFT's is eight times smaller:
// AUTO then keeps true fp32 transforms, except on the fold kernel, which knows when the level makes the difference and
// recomputes those evaluations exactly (guard_loud, fused_plan.cpp).
bool normalised_chain(const syldet_config_t &c)
{
    return c.n_input_fns > 0 && (c.input_fns[0].kind == SYLDET_FN_L2NORMALIZE || c.input_fns[0].kind == SYLDET_FN_NORMALIZE ||
                                 c.input_fns[0].kind == SYLDET_FN_NORMALIZESTD);
}

int64_t count_frames(const syldet *h, int64_t S)
{
    const int64_t need = (int64_t)h->geom.gap + h->cfg.view.window_length;   // :286-288
    if (S < need) return 0;
    return (S - need) / h->geom.hop + 1;                                     // consume hop per frame :299-302
}
int64_t count_evals(const syldet *h, int64_t S)
{
    const int64_t J = count_frames(h, S);
    const int T = h->cfg.view.time_range;
    return J >= T ? J - T + 1 : 0;                                           // SyllableDetector.swift:164-178
}

// Brackets one kernel launch with events on its own stream when profiling is on.
struct KernelTimer {
    syldet *h;
    hipStream_t stream;
    int slot = -1;
    KernelTimer(syldet *h_, hipStream_t s, const char *name) : h(h_), stream(s)
    {
        if (!h->profiling || h->prof_calls.empty()) return;
        syldet::ProfCall &pc = h->prof_calls[(size_t)h->prof_cur()];
        if (pc.count >= syldet::kMaxTimed) return;
        slot = pc.count++;
        pc.names[slot] = name;
        base = (size_t)(h->prof_cur() * syldet::kMaxTimed + slot) * 2;
        if (h->events.size() < (size_t)h->prof_depth * syldet::kMaxTimed * 2) h->events.resize((size_t)h->prof_depth * syldet::kMaxTimed * 2, nullptr);
        for (int k = 0; k < 2; k++)
            if (!h->events[base + (size_t)k]) (void)hipEventCreate(&h->events[base + (size_t)k]);
        (void)hipEventRecord(h->events[base], stream);
    }
    ~KernelTimer()
    {
        if (slot >= 0) (void)hipEventRecord(h->events[base + 1], stream);
    }
    size_t base = 0;
};

// The exact recomputation behind a fused kernel: on ordinary audio an empty launch behind EVERY batch call, so it gets no events of
// its own (two event records are ~9 us on the stream: 1 % of the headline's step) -- it times itself (the device's 100 MHz counter,
// first workgroup in to last workgroup out) and leaves its work list's length and its duration in two page-locked words of the
// call's slot; syldet_timings lists it for the calls that gave it work (a loud recording through a network without a normaliser can
// spend ten times the fused kernel's time here: MEASUREMENTS R5.7).
const char kFixupName[] = "fixup_kernel";
template <class F>
int timed_fixup(syldet *h, hipStream_t stream, FixList list, F launch)
{
    if (h->profiling && !h->prof_calls.empty() && list.counters) {
        if (h->prof_items_n < h->prof_depth) {
            // (the first profiled call since the history's depth grew: once, and nothing may still be writing the old array)
            SYLDET_HIP(hipDeviceSynchronize());
            if (h->prof_items) (void)hipHostFree(h->prof_items);
            h->prof_items = nullptr; h->prof_items_dev = nullptr; h->prof_items_n = 0;
            SYLDET_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->prof_items), 2 * sizeof(unsigned) * (size_t)h->prof_depth, hipHostMallocMapped));
            SYLDET_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->prof_items_dev), h->prof_items, 0));
            h->prof_items_n = h->prof_depth;
            for (int i = 0; i < 2 * h->prof_items_n; i++) h->prof_items[i] = 0u;
        }
        syldet::ProfCall &pc = h->prof_calls[(size_t)h->prof_cur()];
        pc.fix = true;
        pc.fix_stream = stream;
        list.host_count = h->prof_items_dev + 2 * h->prof_cur();
    }
    SYLDET_HIP(launch(list));
    return SYLDET_OK;
}

// samples -> [C][J][F] columns: the fused engine's DFT half where its shape allows (and the handle was not created
// for the generic engine outright), the generic FFT otherwise
int stft_on_stream(syldet *h, const float *d_samples, int64_t stride, int C, int64_t J, float *d_columns, hipStream_t stream,
                   bool for_network = true)
{
    // columns that go through log / dB keep the generic FFT: its error is relative to the frame, the block-floating-point
    // DFT's to the loudest sample of the 128-frame pass, and the logarithm turns relative error of weak bins into absolute
    const bool log_input = for_network && h->cfg.view.scaling != SYLDET_SCALING_LINEAR;
    // ... and columns that meet a network without a normaliser in front keep it too (normalised_chain above); the wide engine
    // is bf16 behind either
    const bool level_input = for_network && !normalised_chain(h->cfg.view) && h->engine != SYLDET_ENGINE_WIDE_BF16 && h->engine_asked == SYLDET_ENGINE_AUTO;
    if (h->has_dft && !log_input && !level_input && (uint64_t)J * (uint64_t)h->geom.bins * 4u < 0xFFFFFFF0ull) {
        FusedDesc d = h->dft.desc;
        fused_segmentation(d, J, C);
        d.spect_out = d_columns;
        d.stamps = nullptr;
        d.ko = 0;
        d.no_fold2 = h->sw.fused_nofold2 ? 1 : 0;
        d.no_cs8 = h->sw.fused_pad128 ? 1 : 0;
        // the fold kernel's spectrogram instantiation where the plan allows it (rows under 2 GiB: its 32-bit byte offsets)
        const bool fold = !h->sw.fused_nofold && fused_s_spectrogram_applicable(d) &&
                          ((J - 1) * (int64_t)d.hop + d.gap + d.W) * 4 < 0x7fffffffll;
        if (!fold && !d.classic_ok) goto generic_transform;
        // the precision guard's work list, and behind the kernel the exact recomputation of the frames it reports
        if (int st = prepare_fix(h, C, J, (J + d.seg_evals - 1) / d.seg_evals + (d.s_seg_evals > 0 ? (J + d.s_seg_evals - 1) / d.s_seg_evals : 0), stream, d.fix)) return st;
        if (fold) {
            KernelTimer t(h, stream, "fused_s_kernel (spectrogram)");
            SYLDET_HIP(launch_fused_s_spectrogram(d, d_samples, stride, C, J, stream));
        } else {
            KernelTimer t(h, stream, "fused_kernel (spectrogram)");
            SYLDET_HIP(launch_fused_spectrogram(d, d_samples, stride, C, J, stream));
        }
        return timed_fixup(h, stream, d.fix, [&](const FixList &l) { return launch_fixup(h->fixd, h->net, d_samples, stride, J, 0, nullptr, nullptr, d_columns, l, stream); });
    }
generic_transform:
    if (!h->sw.no_stft_lanes && stft_lanes_applicable(h->stft, d_samples, stride)) {
        KernelTimer t(h, stream, "stft_lanes_kernel");
        SYLDET_HIP(launch_stft_lanes(h->stft, d_samples, stride, C, J, d_columns, stream));
        return SYLDET_OK;
    }
    KernelTimer t(h, stream, "stft_generic_kernel");
    SYLDET_HIP(launch_stft_generic(h->stft, d_samples, stride, C, J, d_columns, stream));
    return SYLDET_OK;
}

int run_on_stream(syldet *h, const float *d_samples, int64_t S, int64_t stride, int C, float *d_outputs,
                  uint8_t *d_flags, hipStream_t stream)
{
    const int64_t J = count_frames(h, S), E = count_evals(h, S);
    if (E <= 0) return SYLDET_OK;
    SYLDET_HIP(hipSetDevice(h->device));
    h->prof_begin();
    // the fused kernel addresses a channel's results with 32-bit byte offsets; longer rows take the generic engine
    bool fused_route = h->engine == SYLDET_ENGINE_FUSED && (uint64_t)E * (uint64_t)h->geom.outputs * 4u < 0xFFFFFFF0ull;
    FusedDesc d{};
    if (fused_route) {
        d = h->fused.desc;
        fused_segmentation(d, E, C);
        d.stamps = nullptr;
        d.fix = FixList{nullptr, nullptr, 0};
        // diagnostic only: SYLDET_FUSED_KO=<mask> runs an instantiation with parts of the kernel knocked out
        d.ko = h->sw.fused_ko;
        d.force_classic = h->sw.fused_classic ? 1 : 0;
        d.no_fold = h->sw.fused_nofold ? 1 : 0;
        d.no_fold2 = h->sw.fused_nofold2 ? 1 : 0;
        d.no_cs8 = h->sw.fused_pad128 ? 1 : 0;
        // Which fused kernel THIS batch gets is known only now (the fold kernel addresses a row with 32-bit byte offsets; a
        // diagnostic switch may rule it out).  The batch takes the fused route only if that kernel can run it (plans whose hop
        // only the fold kernel holds have no 8-wave form: classic_ok == 0) and, for log / dB columns under AUTO, only on the fold
        // kernel -- the one AUTO committed to for them at create time: the pass-scaled kernels do not hold 1e-5 there.  Anything
        // else goes to the generic engine below, like the long rows.
        const int choice = fused_choice(d, J);
        const bool runnable = choice != 0 || d.classic_ok;
        const bool in_contract = ((h->cfg.view.scaling == SYLDET_SCALING_LINEAR && normalised_chain(h->cfg.view)) || choice == 2 || h->engine_asked == SYLDET_ENGINE_FUSED);
        if (!runnable || !in_contract) fused_route = false;
    }
    if (fused_route) {
        // diagnostic only: SYLDET_FUSED_STAMPS=1 runs the stamped instantiation and prints where a
        // workgroup pass spends its cycles (never set in tests or the benchmark)
        if (h->sw.fused_stamps && fused_s_has_stamps() && fused_choice(d, J) == 2) {
            // the fold kernel's stamped build (-DSYLDET_S_STAMPS): shader clocks its waves spend waiting, summed over all waves
            if (int st = h->d_stamps.reserve(16 * sizeof(unsigned long long))) return st;
            SYLDET_HIP(hipMemsetAsync(h->d_stamps.ptr, 0, 16 * sizeof(unsigned long long), stream));
            d.stamps = (unsigned long long *)h->d_stamps.ptr;
            {
                KernelTimer t(h, stream, "fused_s_kernel");
                SYLDET_HIP(launch_fused(d, d_samples, stride, C, S, J, E, d_outputs, d_flags, stream));
            }
            unsigned long long host[16];
            SYLDET_HIP(hipMemcpyAsync(host, h->d_stamps.ptr, sizeof(host), hipMemcpyDeviceToHost, stream));
            SYLDET_HIP(hipStreamSynchronize(stream));
            const double w = (double)host[4], tl = (double)host[3];
            std::fprintf(stderr, "[syldet stamps] fold kernel: %.0f waves, %.1f tiles each; per tile: %.0f clocks in all, %.0f waiting for the samples' DMA (%.1f %%), "
                                 "%.0f waiting for the LDS reads of the samples (%.1f %%)\n",
                         w, tl / w, (double)host[0] / tl, (double)host[1] / tl, 100.0 * (double)host[1] / (double)host[0], (double)host[2] / tl,
                         100.0 * (double)host[2] / (double)host[0]);
            return SYLDET_OK;
        }
        if (h->sw.fused_stamps) {
            const bool rk = fused_r_applicable(d) && fused_r_has_stamps() && !h->sw.fused_classic;   // which kernel the launcher picks
            const int64_t seg = rk ? d.r_seg_evals : d.seg_evals;
            const size_t n = (size_t)((E + seg - 1) / seg) * (size_t)C * 16;
            if (int st = h->d_stamps.reserve(n * sizeof(unsigned long long))) return st;
            SYLDET_HIP(hipMemsetAsync(h->d_stamps.ptr, 0, n * sizeof(unsigned long long), stream));
            d.stamps = (unsigned long long *)h->d_stamps.ptr;
            {
                KernelTimer t(h, stream, "fused_kernel");
                SYLDET_HIP(launch_fused(d, d_samples, stride, C, S, J, E, d_outputs, d_flags, stream));
            }
            std::vector<unsigned long long> host(n);
            SYLDET_HIP(hipMemcpyAsync(host.data(), h->d_stamps.ptr, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            SYLDET_HIP(hipStreamSynchronize(stream));
            double sum[16] = {0};
            for (size_t i = 0; i < n; i++) sum[i % 16] += (double)host[i];
            double tot = 0;
            for (int i = 0; i < 6; i++) tot += sum[i];            // wave 0's phases add up to the pass; wave 7's are a second view
            const char *names[16] = {"DFT(p) || evaluate(p-1) || block max(p+1)", "barrier 1", "carry + mag + columns", "stage next pass + issue loads",
                                            "barrier 0", "-", "-", "-", "wave 7: DFT || evaluate || block max", "wave 7: barrier 1", "wave 7: carry + mag + columns",
                                            "wave 7: stage + issue loads", "wave 7: barrier 0", "-", "-", "-"};
            if (rk) {
                names[0] = "top of the pass (scale, first fragments)"; names[1] = "ticks 0-23 (stage || evaluate: MFMA phase)";
                names[2] = "ticks 24-47 (stage || evaluate: MFMA phase)"; names[3] = "ticks 48-71 (evaluate: vector phase, magnitudes)";
                names[4] = "ticks 72-95 (magnitudes, block max, strip)"; names[5] = "barrier";
                for (int i = 0; i < 6; i++) names[8 + i] = names[i];
                d.runs = d.r_runs;
            }
            std::fprintf(stderr, "[syldet stamps] runs=%d workgroups=%zu cycles/pass=%.0f\n", d.runs, n / 16, tot / ((double)(n / 16) * d.runs));
            for (int i = 0; i < 16; i++)
                if (sum[i] > 0 && i % 8 < 6) std::fprintf(stderr, "   %-32s %6.0f cycles/pass  %5.1f %%\n", names[i], sum[i] / ((double)(n / 16) * d.runs), 100.0 * sum[i] / tot);
            if (sum[7] > 0)       // (the register-resident-basis kernel's stamped build: whole segments in both clocks)
                std::fprintf(stderr, "   segment: %.0f shader clocks in %.2f us: %.3f GHz\n", sum[6] / (double)(n / 16), sum[7] / (double)(n / 16) * 0.01, sum[6] / (sum[7] * 10.0));
            return SYLDET_OK;
        }
        // the precision guard's work list, and behind the fused kernel the exact recomputation of what it reports
        const int64_t segs = (E + d.r_seg_evals - 1) / d.r_seg_evals + (E + d.seg_evals - 1) / d.seg_evals +
                             (d.s_seg_evals > 0 ? (E + d.s_seg_evals - 1) / d.s_seg_evals : 0);
        if (int st = prepare_fix(h, C, E, segs, stream, d.fix)) return st;
        {
            // (the launcher picks the register-resident-basis kernel where it is instantiated: named for what runs)
            static const char *const names[3] = {"fused_kernel", "fused_r_kernel", "fused_s_kernel"};
            KernelTimer t(h, stream, names[fused_choice(d, J)]);
            SYLDET_HIP(launch_fused(d, d_samples, stride, C, S, J, E, d_outputs, d_flags, stream));
        }
        return timed_fixup(h, stream, d.fix, [&](const FixList &l) { return launch_fixup(h->fixd, h->net, d_samples, stride, J, E, d_outputs, d_flags, nullptr, l, stream); });
    }
    // (+ 16 bytes: the matrix-core network stage reads a frame's last bins as a whole quad)
    if (int st = h->d_columns.reserve((size_t)C * (size_t)J * (size_t)h->geom.bins * sizeof(float) + 16)) return st;
    if (h->engine == SYLDET_ENGINE_WIDE_BF16) {
        if (!h->wide.front)
            if (int st = h->d_xn.reserve((size_t)C * (size_t)E * (size_t)kWideK * 2)) return st;
        if (int st = stft_on_stream(h, d_samples, stride, C, J, (float *)h->d_columns.ptr, stream)) return st;
        if (!h->wide.front) {
            KernelTimer t(h, stream, wide_prep_is_chain(h->net) ? "wide_prep_chain_kernel" : "wide_prep_kernel");
            SYLDET_HIP(launch_wide_prep(h->net, h->geom.bins, (const float *)h->d_columns.ptr, C, J, E, h->d_xn.ptr, stream));
        }
        KernelTimer t(h, stream, h->wide.m32 ? "wide_gemm32s_kernel" : (h->wide.shape16 ? "wide_gemm16_kernel" : "wide_gemm_kernel"));
        SYLDET_HIP(launch_wide_gemm(h->wide, h->d_xn.ptr, (const float *)h->d_columns.ptr, J, E, (int64_t)C * E, d_outputs, d_flags, stream));
        return SYLDET_OK;
    }
    // 1024-point frames in front of a network of the matrix-core class: one launch, the columns never leave the CU
    // frames of four hops: every block transformed once on the matrix cores, then the same network stage -- one launch
    if (h->bdft.ok && !h->sw.no_bdft && (uint64_t)E * 4u < 0xFFFFFFF0ull && (uint64_t)S * 4u < 0x7fffffffull) {
        KernelTimer t(h, stream, "bdft_net_kernel");
        SYLDET_HIP(launch_bdft_net(h->bdft.md, h->bdft.desc, d_samples, stride, C, S, J, E, d_outputs, d_flags, stream));
        return SYLDET_OK;
    }
    if (h->mlpx.ok && !h->sw.no_fft1k && (uint64_t)E * 4u < 0xFFFFFFF0ull &&
        fft1k_applicable(h->stft, h->mlpx.desc, d_samples, stride)) {
        KernelTimer t(h, stream, "fft1k_net_kernel");
        SYLDET_HIP(launch_fft1k_net(h->stft, h->mlpx.desc, d_samples, stride, C, J, E, d_outputs, d_flags, stream));
        return SYLDET_OK;
    }
    if (int st = stft_on_stream(h, d_samples, stride, C, J, (float *)h->d_columns.ptr, stream)) return st;
    if (h->mlpx.ok && (uint64_t)E * 4u < 0xFFFFFFF0ull) {
        KernelTimer t(h, stream, "mlp_mfma_kernel");
        SYLDET_HIP(launch_mlpx(h->mlpx.desc, (const float *)h->d_columns.ptr, C, J, E, d_outputs, d_flags, stream));
        return SYLDET_OK;
    }
    {
        KernelTimer t(h, stream, "mlp_generic_kernel");
        SYLDET_HIP(launch_mlp_generic(h->net, h->geom.bins, (const float *)h->d_columns.ptr, C, J, E, d_outputs,
                                      d_flags, stream));
    }
    return SYLDET_OK;
}

int check_batch_args(const syldet *h, const void *samples, int64_t S, int64_t stride)
{
    if (!h) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL handle");
    if (S < 0) return fail(SYLDET_ERR_INVALID_ARGUMENT, "n_samples must be >= 0");
    if (!samples && S > 0) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL samples");
    // the stride only matters between rows; a single channel may pass anything
    if (h->channels > 1 && stride < S) return fail(SYLDET_ERR_INVALID_ARGUMENT, "channel_stride must be >= n_samples");
    return SYLDET_OK;
}

}  // namespace

extern "C" {

int syldet_create(const syldet_config_t *cfg, int32_t n_channels, int32_t device, int32_t engine, syldet_t **out)
{
    if (!cfg || !out) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (n_channels <= 0 || n_channels > 65535)
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "n_channels must be in [1, 65535]");
    if (engine != SYLDET_ENGINE_AUTO && engine != SYLDET_ENGINE_GENERIC && engine != SYLDET_ENGINE_FUSED && engine != SYLDET_ENGINE_WIDE_BF16)
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "unknown engine");
    std::unique_ptr<syldet> h(new (std::nothrow) syldet());
    if (!h) return fail(SYLDET_ERR_OUT_OF_MEMORY, "out of memory");
    if (int st = h->cfg.assign(*cfg)) return st;
    if (int st = compute_geometry(h->cfg.view, &h->geom)) return st;
    h->sw.read();
    if (h->sw.host_chunk > 0) h->host_chunk_bytes = (size_t)h->sw.host_chunk;

    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(SYLDET_ERR_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count is 0"));
    if (device < 0 || device >= n_dev) return fail(SYLDET_ERR_NO_DEVICE, "device index out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(SYLDET_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SYLDET_ERR_NO_DEVICE, std::string("libsyldet is built for gfx950 only, found ") + prop.gcnArchName);
    SYLDET_HIP(hipSetDevice(device));
    h->device = device;
    h->channels = n_channels;
    h->engine_asked = engine;
    h->engine = SYLDET_ENGINE_GENERIC;
    if (engine != SYLDET_ENGINE_GENERIC && engine != SYLDET_ENGINE_WIDE_BF16) {
        // The fused engine hands columns to the first layer as f16 hi + lo pairs.  For linear |X| columns that is below
        // fp32 noise; log / dB values (magnitude up to ~100, 2^-22 relative = a few 1e-5 absolute) can leave the 1e-5
        // parity bar.  AUTO therefore keeps those scalings on the generic engine; the fused one stays available on request.
        // Round 3: the symmetric-fold kernel transforms every frame at its own scale -- a bin's error is relative to its frame,
        // as the generic FFT's is -- so behind l2normalize (what the split of the logarithms loses is lost relative to the vector
        // the network sees, as in the generic engine's matrix-core network stage) AUTO takes it for log / dB columns too.
        const bool strict_ok = h->cfg.view.scaling == SYLDET_SCALING_LINEAR;
        if (engine == SYLDET_ENGINE_AUTO && !strict_ok) {
            if (make_fused_plan(h->cfg.view, h->geom, h->fused) && fused_s_applicable(h->fused.desc) && h->fused.desc.norm == 1 &&
                !h->sw.fused_nofold && !h->sw.fused_classic) {
                h->engine = SYLDET_ENGINE_FUSED;
            } else {
                h->fused = FusedPlan();
                h->fused.reason = "log/dB scaling: AUTO keeps the generic engine for 1e-5 parity";
            }
        } else if (make_fused_plan(h->cfg.view, h->geom, h->fused)) {
            const bool fold = fused_s_applicable(h->fused.desc) && !h->sw.fused_nofold && !h->sw.fused_classic;
            if (engine == SYLDET_ENGINE_AUTO && !normalised_chain(h->cfg.view) && !fold) {
                // no normaliser in front of the network: the pass-scaled kernels' error follows the recording's level
                // (normalised_chain above); only the fold kernel guards against that
                h->fused = FusedPlan();
                h->fused.reason = "no normaliser in front of the network: AUTO keeps fp32 transforms outside the fold kernel's class";
            } else {
                h->engine = SYLDET_ENGINE_FUSED;
            }
        } else if (engine == SYLDET_ENGINE_FUSED) {
            return fail(SYLDET_ERR_UNSUPPORTED, "fused engine not available for this configuration: " + h->fused.reason);
        }
    }
    h->geom.engine = h->engine;
    SYLDET_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    if (int st = build_tables(h.get())) {
        syldet_destroy(h.release());
        return st;
    }
    if (h->engine == SYLDET_ENGINE_FUSED) {
        int st = upload_fused(h.get());
        if (!st) st = upload_fix(h.get());
        if (st) {
            syldet_destroy(h.release());
            return st;
        }
    }
    // AUTO on the generic engine: the network stage goes to the matrix cores where the configuration is of that kernel's
    // class (a handle created for SYLDET_ENGINE_GENERIC keeps the reference's operation order throughout)
    if (engine == SYLDET_ENGINE_AUTO && h->engine == SYLDET_ENGINE_GENERIC && !h->sw.no_mlpx && make_mlpx_plan(h->cfg.view, h->geom, h->mlpx)) {
        if (int st = upload_mlpx(h.get())) {
            syldet_destroy(h.release());
            return st;
        }
        if (make_bdft_plan(h->cfg.view, h->geom, h->mlpx, h->bdft)) {
            if (int st = upload_bdft(h.get())) {
                syldet_destroy(h.release());
                return st;
            }
        }
    }
    if (engine != SYLDET_ENGINE_GENERIC) {
        int st = build_dft_plan(h.get());
        if (!st && h->has_dft && !h->has_fix) st = upload_fix(h.get());
        if (st) {
            syldet_destroy(h.release());
            return st;
        }
    }
    if (engine == SYLDET_ENGINE_WIDE_BF16) {
        std::string why;
        if (int st = upload_wide(h.get(), why)) {
            syldet_destroy(h.release());
            return why.empty() ? st : fail(st, "wide-network engine not available for this configuration: " + why);
        }
        h->engine = SYLDET_ENGINE_WIDE_BF16;
        h->geom.engine = h->engine;
    }
    // the reference's sample ring (409 600 bytes) + what its feature ring would still hold as columns
    uint64_t need = (uint64_t)(kSampleRingBytes / 4) + (uint64_t)h->geom.gap + (uint64_t)h->cfg.view.window_length +
                    (uint64_t)(h->cfg.view.time_range + 1) * (uint64_t)h->geom.hop, cap = 1;
    while (cap < need) cap <<= 1;
    try {
        h->streams.resize((size_t)n_channels);
        for (auto &s : h->streams) {
            s.reset(new ChannelStream());
            s->last.assign((size_t)h->geom.outputs, 0.0f);   // lastOutputs zeros, SyllableDetector.swift:70
            s->mask = cap - 1;
        }
    } catch (const std::bad_alloc &) {
        syldet_destroy(h.release());
        return fail(SYLDET_ERR_OUT_OF_MEMORY, "out of memory");
    }
    *out = h.release();
    return SYLDET_OK;
}

int syldet_destroy(syldet_t *h)
{
    if (!h) return SYLDET_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamDestroy(h->stream);
    }
    for (hipEvent_t e : h->events)
        if (e) (void)hipEventDestroy(e);
    if (h->prof_items) (void)hipHostFree(h->prof_items);
    for (DeviceBuffer *b : {&h->d_window, &h->d_tw, &h->d_sw, &h->d_params, &h->d_thr, &h->d_columns, &h->d_fused, &h->d_mlpx, &h->d_stamps, &h->d_fix, &h->d_ctab, &h->d_planar, &h->d_wide, &h->d_xn, &h->d_dft, &h->d_stage_in,
                            &h->d_stage_out, &h->d_stage_flags, &h->d_stage_idx, &h->d_stage_cnt})
        b->release();
    h->p_stage_in.release();
    h->p_stage_out.release();
    for (int b = 0; b < 2; b++) {
        h->pipe.d_in[b].release(); h->pipe.d_out[b].release(); h->pipe.d_fl[b].release();
        for (hipEvent_t e : {h->pipe.ev_h2d[b], h->pipe.ev_k[b], h->pipe.ev_d2h[b]})
            if (e) (void)hipEventDestroy(e);
    }
    if (h->pipe.s_in) (void)hipStreamDestroy(h->pipe.s_in);
    if (h->pipe.s_out) (void)hipStreamDestroy(h->pipe.s_out);
    delete h;
    return SYLDET_OK;
}

int syldet_get_geometry(const syldet_t *h, syldet_geometry_t *out)
{
    if (!h || !out) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = h->geom;
    return SYLDET_OK;
}

int32_t syldet_channels(const syldet_t *h) { return h ? h->channels : 0; }

int syldet_profile(syldet_t *h, int enable)
{
    if (!h) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL argument");
    h->profiling = enable != 0;
    h->prof_seq = 0;
    h->prof_calls.clear();
    return SYLDET_OK;
}

int syldet_profile_history(syldet_t *h, int32_t calls)
{
    if (!h || calls < 1 || calls > 65536) return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (hipEvent_t e : h->events)
        if (e) (void)hipEventDestroy(e);
    h->events.clear();
    h->prof_calls.clear();
    h->prof_depth = calls;
    h->prof_seq = 0;
    return SYLDET_OK;
}

int syldet_timings(syldet_t *h, int32_t calls_back, double *milliseconds, const char **names, int32_t capacity, int32_t *count)
{
    if (!h || !count || capacity < 0 || (capacity > 0 && !milliseconds) || calls_back < 0)
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    *count = 0;
    if (h->prof_calls.empty() || calls_back >= h->prof_depth || (int64_t)calls_back >= h->prof_seq) return SYLDET_OK;
    const int idx = (int)((h->prof_seq - 1 - calls_back) % h->prof_depth);
    const syldet::ProfCall &pc = h->prof_calls[(size_t)idx];
    int n = 0;
    for (int i = 0; i < pc.count; i++) {
        const size_t base = (size_t)(idx * syldet::kMaxTimed + i) * 2;
        SYLDET_HIP(hipEventSynchronize(h->events[base + 1]));
        if (n < capacity) {
            float ms = 0.0f;
            SYLDET_HIP(hipEventElapsedTime(&ms, h->events[base], h->events[base + 1]));
            milliseconds[n] = (double)ms;
            if (names) names[n] = pc.names[i];
        }
        n++;
    }
    // the exact path, behind the call's kernels: listed for calls that gave it work, with the duration it measured itself (timed_fixup)
    if (pc.fix && h->prof_items && idx < h->prof_items_n) {
        if (hipStreamSynchronize(pc.fix_stream) != hipSuccess) {       // (a caller's stream that is gone by now: everything, then)
            (void)hipGetLastError();
            SYLDET_HIP(hipDeviceSynchronize());
        }
        if (h->prof_items[2 * idx] != 0u) {
            if (n < capacity) {
                milliseconds[n] = (double)h->prof_items[2 * idx + 1] * 1e-5;
                if (names) names[n] = kFixupName;
            }
            n++;
        }
    }
    *count = n;
    return SYLDET_OK;
}

int syldet_last_timings(syldet_t *h, double *milliseconds, const char **names, int32_t capacity, int32_t *count)
{
    return syldet_timings(h, 0, milliseconds, names, capacity, count);
}

int syldet_fixup_stats(syldet_t *h, int64_t *items, int32_t *overflow)
{
    if (!h) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL argument");
    unsigned c[4] = {0, 0, 0, 0};
    if (h->d_fix.ptr) {
        SYLDET_HIP(hipSetDevice(h->device));
        SYLDET_HIP(hipMemcpy(c, h->d_fix.ptr, sizeof(c), hipMemcpyDeviceToHost));
    }
    if (items) *items = (int64_t)c[2];
    if (overflow) *overflow = (int32_t)c[3];
    return SYLDET_OK;
}

int64_t syldet_segment_evals(const syldet_t *h, int64_t n_samples)
{
    if (!h || h->engine != SYLDET_ENGINE_FUSED) return 0;
    const int64_t E = count_evals(h, n_samples);
    if (E <= 0) return 0;
    FusedDesc d = h->fused.desc;
    fused_segmentation(d, E, h->channels);
    d.force_classic = h->sw.fused_classic ? 1 : 0;
    d.no_fold = h->sw.fused_nofold ? 1 : 0;
    d.ko = h->sw.fused_ko;
    d.stamps = nullptr;
    const int choice = fused_choice(d, count_frames(h, n_samples));
    return choice == 2 ? d.s_seg_evals : (choice == 1 ? d.r_seg_evals : d.seg_evals);
}

int64_t syldet_count_frames(const syldet_t *h, int64_t n_samples) { return h ? count_frames(h, n_samples) : -1; }
int64_t syldet_count_evals(const syldet_t *h, int64_t n_samples) { return h ? count_evals(h, n_samples) : -1; }

int syldet_run_device(syldet_t *h, const float *d_samples, int64_t n_samples, int64_t channel_stride, float *d_outputs,
                      uint8_t *d_flags, void *hip_stream)
{
    if (int st = check_batch_args(h, d_samples, n_samples, channel_stride)) return st;
    return run_on_stream(h, d_samples, n_samples, channel_stride, h->channels, d_outputs, d_flags, (hipStream_t)hip_stream);
}

int syldet_spectrogram_device(syldet_t *h, const float *d_samples, int64_t n_samples, int64_t channel_stride,
                              float *d_columns, void *hip_stream)
{
    if (int st = check_batch_args(h, d_samples, n_samples, channel_stride)) return st;
    if (!d_columns) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL argument");
    SYLDET_HIP(hipSetDevice(h->device));
    const int64_t J = count_frames(h, n_samples);
    h->prof_begin();
    return stft_on_stream(h, d_samples, channel_stride, h->channels, J, d_columns, (hipStream_t)hip_stream, false);
}

int syldet_detections_device(syldet_t *h, const uint8_t *d_flags, int64_t n_evals, double debounce_seconds,
                             int64_t *d_indices, int64_t capacity, int64_t *d_counts, void *hip_stream)
{
    if (!h || !d_flags || n_evals < 0 || capacity < 0 || (capacity > 0 && !d_indices))
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    SYLDET_HIP(hipSetDevice(h->device));
    const int64_t debounce_frames = (int64_t)(debounce_seconds * h->cfg.view.sampling_rate);   // TrackDetector.swift:19-26
    SYLDET_HIP(launch_detections(d_flags, h->channels, n_evals, h->geom.first_index, h->geom.hop, debounce_frames,
                                 d_indices, capacity, d_counts, (hipStream_t)hip_stream));
    return SYLDET_OK;
}

// ---- host-pointer conveniences: stage, run, copy back, block ----

static int pipe_bring_up(syldet *h)
{
    syldet::Pipe &p = h->pipe;
    if (p.up) return SYLDET_OK;
    if (!p.s_in) SYLDET_HIP(hipStreamCreateWithFlags(&p.s_in, hipStreamNonBlocking));
    if (!p.s_out) SYLDET_HIP(hipStreamCreateWithFlags(&p.s_out, hipStreamNonBlocking));
    for (int b = 0; b < 2; b++)
        for (hipEvent_t *e : {&p.ev_h2d[b], &p.ev_k[b], &p.ev_d2h[b]})
            if (!*e) SYLDET_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    p.up = true;
    return SYLDET_OK;
}

// The batch call on host buffers: TrackDetector's loop reads a recording buffer by buffer and runs the detector on each
// (TrackDetector.swift:45-105); here the recording is cut along time into stages of about host_chunk_bytes of input, each
// stage the evaluations [e0, e0 + n) of every channel with the samples they reach ((n + T - 2) hop + gap + W: the halo of
// (T - 1) hop + W - hop samples is read by both neighbours), and the stages flow through two sets of device buffers: the H2D
// copy of stage k + 1 (copy-in stream) runs under the kernel of stage k (the handle's stream) and the D2H copy of stage k - 1
// (copy-out stream).  Device staging is bounded by two stages, whatever the recording's length.  Kernels whose scales
// are per frame or per hop-aligned block (the fold kernel, the FFT and block-transform engines) give a stage's evaluations
// the bits the one-shot call gives them; the pass-scaled kernels agree to a few 1e-7, as between any two tilings (syldet.h,
// syldet_fixup_stats).  The copies read and write the caller's rows in place: page-locked memory (syldet_host_alloc) is
// truly asynchronous; for ordinary memory the runtime pins the pages of each copy on the fly and the copy call returns when
// its bytes have moved -- the kernel of the stage before runs meanwhile all the same (measured: within a few per cent of
// the PCIe rate either way, and a staging copy through a pinned buffer of our own was half as fast).
int syldet_run(syldet_t *h, const float *samples, int64_t n_samples, int64_t channel_stride, float *outputs, uint8_t *flags)
{
    if (int st = check_batch_args(h, samples, n_samples, channel_stride)) return st;
    std::lock_guard<std::mutex> staging(h->pump_mu);   // the staging buffers and h->stream: one user at a time
    SYLDET_HIP(hipSetDevice(h->device));
    const int C = h->channels, n_out = h->geom.outputs, T = h->cfg.view.time_range;
    const int64_t E = count_evals(h, n_samples), hop = h->geom.hop, frame = (int64_t)h->geom.gap + h->cfg.view.window_length;
    if (E <= 0) return SYLDET_OK;
    if (int st = pipe_bring_up(h)) return st;
    syldet::Pipe &p = h->pipe;

    // evaluations per stage: about host_chunk_bytes of input, the stages of equal size
    const int64_t fixed = (int64_t)C * ((T - 2) * hop + frame) * 4, per_eval = (int64_t)C * hop * 4;
    int64_t n_c = ((int64_t)h->host_chunk_bytes - fixed) / per_eval;
    if (n_c < 1) n_c = 1;
    const int64_t n_stages = (E + n_c - 1) / n_c;
    n_c = (E + n_stages - 1) / n_stages;
    const int64_t S_max = (n_c + T - 2) * hop + frame;
    for (int b = 0; b < (n_stages > 1 ? 2 : 1); b++) {
        if (int st = p.d_in[b].reserve((size_t)C * (size_t)S_max * 4)) return st;
        if (int st = p.d_out[b].reserve((size_t)C * (size_t)n_c * (size_t)n_out * 4)) return st;
        if (int st = p.d_fl[b].reserve((size_t)C * (size_t)n_c)) return st;
    }
    // whatever happens, nothing of this call is in flight when it returns
    struct Drain {
        syldet *h;
        ~Drain()
        {
            (void)hipStreamSynchronize(h->pipe.s_in);
            (void)hipStreamSynchronize(h->stream);
            (void)hipStreamSynchronize(h->pipe.s_out);
        }
    } drain{h};

    for (int64_t k = 0; k < n_stages; k++) {
        const int b = (int)(k & 1);
        const int64_t e0 = k * n_c, n = std::min(n_c, E - e0), s0 = e0 * hop, Sc = (n + T - 2) * hop + frame;
        if (k >= 2) SYLDET_HIP(hipStreamWaitEvent(p.s_in, p.ev_k[b], 0));      // the kernel of stage k - 2 has read this device buffer
        SYLDET_HIP(hipMemcpy2DAsync(p.d_in[b].ptr, (size_t)Sc * 4, samples + s0, (size_t)channel_stride * 4, (size_t)Sc * 4, (size_t)C,
                                    hipMemcpyHostToDevice, p.s_in));
        SYLDET_HIP(hipEventRecord(p.ev_h2d[b], p.s_in));
        SYLDET_HIP(hipStreamWaitEvent(h->stream, p.ev_h2d[b], 0));
        if (k >= 2) SYLDET_HIP(hipStreamWaitEvent(h->stream, p.ev_d2h[b], 0)); // stage k - 2's results have left the device buffers
        if (int st = run_on_stream(h, (const float *)p.d_in[b].ptr, Sc, Sc, C, (float *)p.d_out[b].ptr, (uint8_t *)p.d_fl[b].ptr, h->stream)) return st;
        SYLDET_HIP(hipEventRecord(p.ev_k[b], h->stream));
        SYLDET_HIP(hipStreamWaitEvent(p.s_out, p.ev_k[b], 0));
        if (outputs)
            SYLDET_HIP(hipMemcpy2DAsync(outputs + (size_t)e0 * n_out, (size_t)E * n_out * 4, p.d_out[b].ptr, (size_t)n * n_out * 4,
                                        (size_t)n * n_out * 4, (size_t)C, hipMemcpyDeviceToHost, p.s_out));
        if (flags)
            SYLDET_HIP(hipMemcpy2DAsync(flags + e0, (size_t)E, p.d_fl[b].ptr, (size_t)n, (size_t)n, (size_t)C, hipMemcpyDeviceToHost, p.s_out));
        SYLDET_HIP(hipEventRecord(p.ev_d2h[b], p.s_out));
    }
    SYLDET_HIP(hipStreamSynchronize(p.s_out));
    return SYLDET_OK;
}

// Host memory for audio and results that the device reads and writes in place (page-locked): what the reference's ring
// allocation (TPCircularBufferInit, TPCircularBuffer.c:43-124: the buffer the audio thread produces into) becomes for a
// host that feeds a GPU.  Buffers from here make syldet_run skip its staging copies.
int syldet_host_alloc(size_t bytes, void **out)
{
    if (!out) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes > 0 ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        *out = nullptr;
        return fail(e == hipErrorOutOfMemory ? SYLDET_ERR_OUT_OF_MEMORY : SYLDET_ERR_DEVICE, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    return SYLDET_OK;
}

int syldet_host_free(void *p)
{
    if (!p) return SYLDET_OK;
    SYLDET_HIP(hipHostFree(p));
    return SYLDET_OK;
}

// ---- interleaved (frame-major) audio: de-interleave on the device, then the batch path ----

int syldet_run_interleaved_device(syldet_t *h, const float *d_interleaved, int64_t n_frames, int32_t total_channels,
                                  float *d_outputs, uint8_t *d_flags, void *hip_stream)
{
    if (!h) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL handle");
    if (total_channels != h->channels) return fail(SYLDET_ERR_INVALID_ARGUMENT, "total_channels must equal the bank's channel count");
    if (n_frames < 0) return fail(SYLDET_ERR_INVALID_ARGUMENT, "n_frames must be >= 0");
    if (count_evals(h, n_frames) <= 0) return SYLDET_OK;
    if (!d_interleaved) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL buffer");
    SYLDET_HIP(hipSetDevice(h->device));
    const int C = h->channels;
    if (int st = h->d_planar.reserve((size_t)C * (size_t)n_frames * sizeof(float))) return st;
    SYLDET_HIP(launch_deinterleave(d_interleaved, n_frames, C, 0, C, (float *)h->d_planar.ptr, n_frames, (hipStream_t)hip_stream));
    return run_on_stream(h, (const float *)h->d_planar.ptr, n_frames, n_frames, C, d_outputs, d_flags, (hipStream_t)hip_stream);
}

int syldet_run_interleaved(syldet_t *h, const float *interleaved, int64_t n_frames, int32_t total_channels, float *outputs,
                           uint8_t *flags)
{
    if (!h) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL handle");
    if (total_channels != h->channels) return fail(SYLDET_ERR_INVALID_ARGUMENT, "total_channels must equal the bank's channel count");
    if (n_frames < 0) return fail(SYLDET_ERR_INVALID_ARGUMENT, "n_frames must be >= 0");
    const int C = h->channels;
    const int64_t E = count_evals(h, n_frames);
    if (E <= 0) return SYLDET_OK;
    if (!interleaved) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL buffer");
    std::lock_guard<std::mutex> staging(h->pump_mu);   // the staging buffers and h->stream: one user at a time
    SYLDET_HIP(hipSetDevice(h->device));
    const size_t in_bytes = (size_t)C * (size_t)n_frames * sizeof(float);
    const size_t out_bytes = (size_t)C * (size_t)E * (size_t)h->geom.outputs * sizeof(float);
    const size_t fl_bytes = (size_t)C * (size_t)E;
    if (int st = h->d_stage_in.reserve(in_bytes)) return st;
    if (int st = h->d_stage_out.reserve(out_bytes)) return st;
    if (int st = h->d_stage_flags.reserve(fl_bytes)) return st;
    SYLDET_HIP(hipMemcpyAsync(h->d_stage_in.ptr, interleaved, in_bytes, hipMemcpyHostToDevice, h->stream));
    if (int st = syldet_run_interleaved_device(h, (const float *)h->d_stage_in.ptr, n_frames, total_channels,
                                               (float *)h->d_stage_out.ptr, (uint8_t *)h->d_stage_flags.ptr, h->stream))
        return st;
    if (outputs) SYLDET_HIP(hipMemcpyAsync(outputs, h->d_stage_out.ptr, out_bytes, hipMemcpyDeviceToHost, h->stream));
    if (flags) SYLDET_HIP(hipMemcpyAsync(flags, h->d_stage_flags.ptr, fl_bytes, hipMemcpyDeviceToHost, h->stream));
    SYLDET_HIP(hipStreamSynchronize(h->stream));
    return SYLDET_OK;
}

int syldet_spectrogram(syldet_t *h, const float *samples, int64_t n_samples, int64_t channel_stride, float *columns)
{
    if (int st = check_batch_args(h, samples, n_samples, channel_stride)) return st;
    if (!columns) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> staging(h->pump_mu);   // the staging buffers and h->stream: one user at a time
    SYLDET_HIP(hipSetDevice(h->device));
    const int C = h->channels;
    const int64_t J = count_frames(h, n_samples);
    if (J <= 0) return SYLDET_OK;
    const size_t in_bytes = (size_t)C * (size_t)n_samples * sizeof(float);
    const size_t col_bytes = (size_t)C * (size_t)J * (size_t)h->geom.bins * sizeof(float);
    if (int st = h->d_stage_in.reserve(in_bytes)) return st;
    if (int st = h->d_columns.reserve(col_bytes)) return st;
    SYLDET_HIP(hipMemcpy2DAsync(h->d_stage_in.ptr, (size_t)n_samples * sizeof(float), samples,
                                (size_t)channel_stride * sizeof(float), (size_t)n_samples * sizeof(float), (size_t)C,
                                hipMemcpyHostToDevice, h->stream));
    h->prof_begin();
    if (int st = stft_on_stream(h, (const float *)h->d_stage_in.ptr, n_samples, C, J, (float *)h->d_columns.ptr, h->stream, false)) return st;
    SYLDET_HIP(hipMemcpyAsync(columns, h->d_columns.ptr, col_bytes, hipMemcpyDeviceToHost, h->stream));
    SYLDET_HIP(hipStreamSynchronize(h->stream));
    return SYLDET_OK;
}

int syldet_detections(syldet_t *h, const uint8_t *flags, int64_t n_evals, double debounce_seconds, int64_t *indices,
                      int64_t capacity, int64_t *counts)
{
    if (!h || !flags || !counts || n_evals < 0 || capacity < 0 || (capacity > 0 && !indices))
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    std::lock_guard<std::mutex> staging(h->pump_mu);   // the staging buffers and h->stream: one user at a time
    SYLDET_HIP(hipSetDevice(h->device));
    const int C = h->channels;
    const size_t fl_bytes = std::max<size_t>((size_t)C * (size_t)n_evals, 1);
    const size_t idx_bytes = std::max<size_t>((size_t)C * (size_t)capacity * sizeof(int64_t), 8);
    if (int st = h->d_stage_flags.reserve(fl_bytes)) return st;
    if (int st = h->d_stage_idx.reserve(idx_bytes)) return st;
    if (int st = h->d_stage_cnt.reserve((size_t)C * sizeof(int64_t))) return st;
    if (n_evals > 0)
        SYLDET_HIP(hipMemcpyAsync(h->d_stage_flags.ptr, flags, (size_t)C * (size_t)n_evals, hipMemcpyHostToDevice, h->stream));
    if (int st = syldet_detections_device(h, (const uint8_t *)h->d_stage_flags.ptr, n_evals, debounce_seconds,
                                          capacity > 0 ? (int64_t *)h->d_stage_idx.ptr : nullptr, capacity,
                                          (int64_t *)h->d_stage_cnt.ptr, h->stream))
        return st;
    if (capacity > 0)
        SYLDET_HIP(hipMemcpyAsync(indices, h->d_stage_idx.ptr, (size_t)C * (size_t)capacity * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    SYLDET_HIP(hipMemcpyAsync(counts, h->d_stage_cnt.ptr, (size_t)C * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    SYLDET_HIP(hipStreamSynchronize(h->stream));
    return SYLDET_OK;
}

// ---- streaming front-end -------------------------------------------------------------
// The reference keeps, per detector, a 409600-byte sample ring drained one frame at a
// time and a feature ring of F-float columns drained one column per evaluation.  Here a
// channel keeps the raw samples from the first frame of its next evaluation onward; when
// the consumer asks for a value and none is queued, every evaluation those samples allow
// is computed in one device pass and queued.  Results are the batch engine's.

int syldet_append(syldet_t *h, int32_t channel, const float *data, int64_t n_samples)
{
    if (!h || channel < 0 || channel >= h->channels || n_samples < 0 || (n_samples > 0 && !data))
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    ChannelStream &cs = *h->streams[(size_t)channel];
    if (!cs.has_room(n_samples, h->geom.hop)) return fail(SYLDET_ERR_BUFFER_FULL, "Insufficient space on buffer.");
    if (!cs.ensure_ring()) return fail(SYLDET_ERR_OUT_OF_MEMORY, "out of memory");
    cs.write(data, n_samples, 1);
    return SYLDET_OK;
}

int syldet_append_interleaved(syldet_t *h, const float *data, int64_t n_frames, int32_t total_channels)
{
    if (!h || n_frames < 0 || (n_frames > 0 && !data) || total_channels != h->channels)
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    // all channels or none: a caller that retries after "buffer full" must not double a block on some of them
    // (room only grows between the check and the writes: the consumer is the only other party)
    for (int c = 0; c < h->channels; c++)
        if (!h->streams[(size_t)c]->has_room(n_frames, h->geom.hop)) return fail(SYLDET_ERR_BUFFER_FULL, "Insufficient space on buffer.");
    for (int c = 0; c < h->channels; c++)
        if (!h->streams[(size_t)c]->ensure_ring()) return fail(SYLDET_ERR_OUT_OF_MEMORY, "out of memory");
    for (int c = 0; c < h->channels; c++) h->streams[(size_t)c]->write(data + c, n_frames, total_channels);
    return SYLDET_OK;
}

int syldet_append_interleaved_channels(syldet_t *h, const float *data, int64_t n_frames, int32_t total_channels, const int32_t *source_channel)
{
    if (!h || n_frames < 0 || (n_frames > 0 && !data) || total_channels < 1 || !source_channel)
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    for (int c = 0; c < h->channels; c++)
        if (source_channel[c] < 0 || source_channel[c] >= total_channels) return fail(SYLDET_ERR_INVALID_ARGUMENT, "source channel outside the stream");
    // all channels or none, as syldet_append_interleaved
    for (int c = 0; c < h->channels; c++)
        if (!h->streams[(size_t)c]->has_room(n_frames, h->geom.hop)) return fail(SYLDET_ERR_BUFFER_FULL, "Insufficient space on buffer.");
    for (int c = 0; c < h->channels; c++)
        if (!h->streams[(size_t)c]->ensure_ring()) return fail(SYLDET_ERR_OUT_OF_MEMORY, "out of memory");
    for (int c = 0; c < h->channels; c++) h->streams[(size_t)c]->write(data + source_channel[c], n_frames, total_channels);
    return SYLDET_OK;
}

// Evaluates what the listed channels have pending: channels with the same number of new evaluations share one
// pinned staging block, one H2D copy, one launch and one D2H copy; the results join each channel's queue of
// computed evaluations.  Adds the number queued to *queued.
static int pump_impl(syldet *h, const int32_t *channels, int32_t n, int64_t *queued)
{
    std::lock_guard<std::mutex> pump_lock(h->pump_mu);
    const int n_out = h->geom.outputs;
    const int64_t hop = h->geom.hop, frame = (int64_t)h->geom.gap + h->cfg.view.window_length;
    std::map<int64_t, std::vector<int32_t>> groups;          // evaluations -> channels
    for (int32_t i = 0; i < n; i++) {
        ChannelStream &cs = *h->streams[(size_t)channels[i]];
        const uint64_t tail = cs.tail.load(std::memory_order_acquire);
        // `while processFourierData() {}` (SyllableDetector.swift:155): every whole frame is extracted now
        cs.frames_done.store(count_frames(h, (int64_t)tail), std::memory_order_release);
        const int64_t E = count_evals(h, (int64_t)(tail - cs.head.load(std::memory_order_relaxed)));
        if (E > 0) groups[E].push_back(channels[i]);
    }
    if (groups.empty()) return SYLDET_OK;
    SYLDET_HIP(hipSetDevice(h->device));
    for (auto &g : groups) {
        const int64_t E = g.first, S = frame + (E + h->cfg.view.time_range - 2) * hop;   // samples E evaluations span
        const size_t nc = g.second.size();
        if (int st = h->p_stage_in.reserve(nc * (size_t)S * sizeof(float))) return st;
        if (int st = h->p_stage_out.reserve(nc * (size_t)E * (size_t)n_out * sizeof(float))) return st;
        if (int st = h->d_stage_in.reserve(nc * (size_t)S * sizeof(float))) return st;
        if (int st = h->d_stage_out.reserve(nc * (size_t)E * (size_t)n_out * sizeof(float))) return st;
        float *in = (float *)h->p_stage_in.ptr, *outs = (float *)h->p_stage_out.ptr;
        for (size_t k = 0; k < nc; k++) {
            const ChannelStream &cs = *h->streams[(size_t)g.second[k]];
            cs.copy_out(cs.head.load(std::memory_order_relaxed), in + k * (size_t)S, (size_t)S);
        }
        SYLDET_HIP(hipMemcpyAsync(h->d_stage_in.ptr, in, nc * (size_t)S * sizeof(float), hipMemcpyHostToDevice, h->stream));
        if (int st = run_on_stream(h, (const float *)h->d_stage_in.ptr, S, S, (int)nc, (float *)h->d_stage_out.ptr, nullptr, h->stream))
            return st;
        SYLDET_HIP(hipMemcpyAsync(outs, h->d_stage_out.ptr, nc * (size_t)E * (size_t)n_out * sizeof(float), hipMemcpyDeviceToHost,
                                  h->stream));
        SYLDET_HIP(hipStreamSynchronize(h->stream));
        for (size_t k = 0; k < nc; k++) {
            ChannelStream &cs = *h->streams[(size_t)g.second[k]];
            const float *o = outs + k * (size_t)E * (size_t)n_out;
            {
                std::lock_guard<std::mutex> lock(cs.mu);
                for (int64_t e = 0; e < E; e++) cs.ready.emplace_back(o + (size_t)e * (size_t)n_out, o + (size_t)(e + 1) * (size_t)n_out);
            }
            // each evaluation consumes one column = hop samples (:175-178)
            cs.head.store(cs.head.load(std::memory_order_relaxed) + (uint64_t)(E * hop), std::memory_order_release);
            if (queued) *queued += E;
        }
    }
    return SYLDET_OK;
}

static int pump(syldet *h, const int32_t *channels, int32_t n, int64_t *queued)
{
    try {                                                        // no exception crosses the C boundary
        return pump_impl(h, channels, n, queued);
    } catch (const std::bad_alloc &) {
        return fail(SYLDET_ERR_OUT_OF_MEMORY, "out of memory");
    }
}

int syldet_process_new_value(syldet_t *h, int32_t channel)
{
    if (!h || channel < 0 || channel >= h->channels) return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    ChannelStream &cs = *h->streams[(size_t)channel];
    if (int st = pump(h, &channel, 1, nullptr)) return st;
    std::lock_guard<std::mutex> lock(cs.mu);
    if (cs.ready.empty()) return 0;
    cs.last = std::move(cs.ready.front());
    cs.ready.pop_front();
    return 1;
}

int syldet_process_all(syldet_t *h, int64_t *n_queued)
{
    if (!h) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL argument");
    std::vector<int32_t> all;
    try {
        all.resize((size_t)h->channels);
    } catch (const std::bad_alloc &) {
        return fail(SYLDET_ERR_OUT_OF_MEMORY, "out of memory");
    }
    for (int32_t c = 0; c < h->channels; c++) all[(size_t)c] = c;
    int64_t queued = 0;
    if (int st = pump(h, all.data(), h->channels, &queued)) return st;
    if (n_queued) *n_queued = queued;
    return SYLDET_OK;
}

int64_t syldet_pending_evaluations(const syldet_t *h, int32_t channel)
{
    if (!h || channel < 0 || channel >= h->channels) return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    ChannelStream &cs = *h->streams[(size_t)channel];
    std::lock_guard<std::mutex> lock(cs.mu);
    return (int64_t)cs.ready.size();
}

int syldet_last_outputs(const syldet_t *h, int32_t channel, float *out)
{
    if (!h || !out || channel < 0 || channel >= h->channels) return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    ChannelStream &cs = *h->streams[(size_t)channel];
    std::lock_guard<std::mutex> lock(cs.mu);
    std::copy(cs.last.begin(), cs.last.end(), out);
    return SYLDET_OK;
}

int syldet_last_detected(const syldet_t *h, int32_t channel)
{
    if (!h || channel < 0 || channel >= h->channels) return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    ChannelStream &cs = *h->streams[(size_t)channel];
    std::lock_guard<std::mutex> lock(cs.mu);
    return (double)cs.last[0] >= h->cfg.view.thresholds[0] ? 1 : 0;   // SyllableDetector.swift:27-31
}

int syldet_seen_syllable(syldet_t *h, int32_t channel)
{
    int ret = 0;                                                       // SyllableDetector.swift:220-230
    for (;;) {
        const int r = syldet_process_new_value(h, channel);
        if (r < 0) return r;
        if (r == 0) break;
        if (syldet_last_detected(h, channel) == 1) ret = 1;
    }
    return ret;
}

}  // extern "C"
