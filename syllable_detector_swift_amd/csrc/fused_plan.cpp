// fused_plan.cpp -- decides whether a configuration fits the fused engine and, if so, builds its
// constant tables on the host in fp64: the split-f16 DFT basis fragments, the first layer folded with
// the affine input maps, and the LDS layout.
//
// Folding (reference: NeuralNet.apply, Common/NeuralNet.swift:294-326):
//   input chain  = [one of l2normalize | normalize | normalizestd]?  then  (mapminmax | mapstd)*
//   the affine tail composes to  x = a o v' + b  (per position);  the head is  v' = alpha*v + beta*1
//   with per-window scalars (alpha, beta) = (1/||v||, 0) | (2/(mx-mn), (-mn-mx)/(mx-mn)) | (1/sigma, -mu/sigma);
//   layer 0:  W0.x + b0 = alpha * (W0 o a).v + beta * (W0 o a).1 + (b0 + W0.b)
//   and (W0 o a).v = sum_t (W0 o a)_t . column(e + t) is a GEMM over the T taps of the window whose B
//   operand is the column buffer at row offset e + t (kernels_fused.hip).
// Anything else (a normaliser after an affine map, three or more layers, wide layers, long windows)
// runs on the generic engine.

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "fused_plan.hpp"

namespace sd {

namespace {

// float -> IEEE binary16, round to nearest even
uint16_t to_half(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t man = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
    if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - exp;
        uint32_t h = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (h & 1))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)exp << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t)(sign | h);
}

float from_half(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ffu;
    uint32_t x;
    if (exp == 0) {
        if (man == 0) x = sign;
        else {
            int e = -1;
            uint32_t m = man;
            do { e++; m <<= 1; } while (!(m & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (exp == 31) x = sign | 0x7f800000u | (man << 13);
    else x = sign | ((exp - 15 + 127) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

void split_half(double v, uint16_t &hi, uint16_t &lo)
{
    hi = to_half((float)v);
    lo = to_half((float)(v - (double)from_half(hi)));
}

}  // namespace

bool make_fused_plan(const syldet_config_t &c, const syldet_geometry_t &g, FusedPlan &p)
{
    p = FusedPlan();
    auto no = [&p](const char *why) { p.reason = why; return false; };
    const int N = c.fourier_length, W = c.window_length, F = g.bins, T = c.time_range;
    // |X|^2 columns (extractMagnitude, CircularShortTimeFourierTransform.swift:221-278; never used by the detector) square the
    // dynamic range: one f16 scale per pass cannot hold a quiet pass next to a loud one.  Generic engine.
    if (c.spectrum == SYLDET_SPECTRUM_MAGNITUDE) return no("|X|^2 columns");
    if (g.hop % 4 != 0) return no("hop is not a multiple of 4");
    if (W > 256) return no("window longer than 256 samples");
    if (W % 4 != 0) return no("window length is not a multiple of 4");
    if (c.n_layers < 1 || c.n_layers > 2) return no("more than two layers");
    const syldet_layer_t &L0 = c.layers[0];
    const int H = L0.outputs;
    const int n_out = g.outputs;
    if (c.n_layers == 2 && n_out > 4) return no("more than 4 outputs");
    if (c.n_output_fns > kMaxFns) return no("too many output functions");
    const int KS = W <= 128 ? 4 : 8;              // k-steps of 32 samples
    if (fused_taps_max(T) == 0) return no("timeRange above 12");

    // ---- input chain pattern
    const int I = g.inputs;
    int norm = 0, first_affine = 0;
    if (c.n_input_fns > 0) {
        const int k0 = c.input_fns[0].kind;
        if (k0 == SYLDET_FN_L2NORMALIZE) { norm = 1; first_affine = 1; }
        else if (k0 == SYLDET_FN_NORMALIZE) { norm = 2; first_affine = 1; }
        else if (k0 == SYLDET_FN_NORMALIZESTD) { norm = 3; first_affine = 1; }
    }
    // more than 32 bins: only the fold kernel's twice-folded form with two row tiles per parity holds them (W == N == 256, one quad
    // of hidden units, plain ring: kernels_fused_s.hip, NT = 2); the other fused kernels and tables below are built for 32
    const bool wide_band = F > 32;
    if (F > 64 || (wide_band && !(W == 256 && N == 256 && H <= 4 && c.n_layers == 2 && g.hop % 64 != 0))) return no("too many bins");
    if (H > 16) return no("first layer too wide");
    std::vector<double> a((size_t)I, 1.0), b((size_t)I, 0.0);
    for (int k = first_affine; k < c.n_input_fns; k++) {
        const syldet_fn_t &f = c.input_fns[k];
        if (f.kind != SYLDET_FN_MAPMINMAX && f.kind != SYLDET_FN_MAPSTD) return no("a normaliser follows another input function");
        for (int i = 0; i < I; i++) {          // x <- (x - xoff) * gain + y   (MapMinMax.apply :127-131, MapStd.apply :162-169)
            a[(size_t)i] = a[(size_t)i] * (double)f.gains[i];
            b[(size_t)i] = (b[(size_t)i] - (double)f.x_offsets[i]) * (double)f.gains[i] + (double)f.y;
        }
    }

    // ---- geometry of a pass
    const int hop = g.hop;
    const int nsmp = (kFusedTileFrames - 1) * hop + KS * 32;    // samples one pass's frames read
    const int nload = (nsmp / 4 + kFusedBlock - 1) / kFusedBlock;
    const bool classic_hop_ok = nload <= kFusedMaxLoads;   // (else only the symmetric-fold kernel, with its own ring geometry, may take the shape)
    // LDS bank spreading: a lane reads 8 consecutive f16 samples of its frame with two ds_read_b64; the 16
    // frames of a tile are hop/2 dwords apart, which spreads over the 64 banks unless hop is a multiple of 16
    // (hop = 128: every frame on the same bank).  Then 4 halves of padding follow every hop staged samples;
    // 8-sample groups never straddle it because they start at multiples of 8 inside a frame.
    const int skew = (hop % 16 == 0) ? 4 : 0;
    auto skewed = [&](int i) { return i + skew * (i / hop); };
    const int nsmp_p = (skewed(nsmp + 16) + 15) / 8 * 8;
    const int PS = kFusedTileFrames + 2 * (T - 1);   // transition strip + the pass's own columns

    FusedDesc &d = p.desc;
    d.W = W; d.KS = KS; d.hop = hop; d.gap = g.gap; d.F = F; d.T = T; d.H = H; d.norm = norm;
    d.scaling = c.scaling;
    d.n_layers = c.n_layers; d.n_out = n_out; d.tf0 = L0.transfer;
    d.tf1 = c.n_layers == 2 ? c.layers[1].transfer : SYLDET_TF_PURELIN;
    d.rule = c.rule; d.n_out_fns = c.n_output_fns; d.I = I;
    d.nsmp = nsmp; d.nload = nload; d.skew = skew;
    d.hop_magic = (unsigned)((0x100000000ull + (unsigned)hop - 1) / (unsigned)hop);
    d.ps = PS; d.smp_stride = nsmp_p;
    int off = 0;
    auto take = [&off](int bytes) { const int o = off; off += (bytes + 15) / 16 * 16; return o; };
    d.lds_dfrag = take(KS * 8 * 1024);
    d.lds_smp = take(2 * nsmp_p * 2);                // staged samples of one pass, f16 hi + lo
    d.lds_colh = take(PS * kFusedColStride * 2);     // |X| columns, f16 hi
    d.lds_coll = take(PS * kFusedColStride * 2);     //              f16 lo
    d.lds_stat = take(2 * PS * 4);
    d.lds_red = take(64);
    d.lds_cst = take((32 + kMaxFns * 33) * 4);       // thresholds + output maps (kCst* in kernels_fused.hip)
    d.lds_total = off;
    d.classic_ok = (off <= 160 * 1024 && classic_hop_ok && !wide_band) ? 1 : 0;        // (the register-resident-basis kernel has its own, smaller layout: decided below)
    {   // the register-resident-basis kernel's own pass geometry and LDS layout
        const int rn = (kFusedRTileFrames - 1) * hop + KS * 32;
        const int rl = (rn / 4 + kFusedRBlock - 1) / kFusedRBlock;
        // every thread always stages all of its quads: room for 1024 rl samples (those past rn are zeros nobody reads)
        // (the instantiated kernel stages 9 quads a thread whatever the hop: room for those)
        // (its own bank spreading: 4 halves behind every hop samples where the hop is a power of two, behind every 16
        // samples for the other multiples of 16 -- kernels_fused_r.hip)
        const bool pow2 = hop == 16 || hop == 32 || hop == 64 || hop == 128;
        auto rskewed = [&](int i) { return skew == 0 ? i : (pow2 ? i + 4 * (i / hop) : i + 4 * (i / 16)); };
        const int rn_p = (rskewed(std::max(rn, 4 * kFusedRBlock * std::max(rl, 9)) + 16) + 15) / 8 * 8;
        const int rps = kFusedRTileFrames + 2 * (T - 1);
        int roff = 0;
        auto rtake = [&roff](int bytes) { const int o = roff; roff += (bytes + 15) / 16 * 16; return o; };
        d.r_nsmp = rn; d.r_nload = rl; d.r_ps = rps; d.r_smp_stride = rn_p;
        d.r_lds_smp = rtake(2 * 2 * rn_p * 2);           // two buffers of f16 hi + lo
        d.r_lds_p = rtake(kFusedRPRows * 52 * 4);        // tap products: a ring of three passes' frames
        d.r_lds_red = rtake(64);
        d.r_lds_cst = rtake((32 + kMaxFns * 33) * 4);
        d.r_lds_total = roff;
        d.r_ok = (rl <= kFusedRMaxLoads && roff <= 160 * 1024 && !wide_band) ? 1 : 0;
    }

    // ---- DFT basis fragments: A operand of v_mfma_f32_16x16x32_f16, lane l holds row l&15 of its tile,
    // k = 8*(l>>4) + j.  Tiles: re bins 0-15, re 16-31, im 0-15, im 16-31.  Basis row r < F:
    // re = w[n] cos(2 pi (f0+r) n / N), im = -w[n] sin(...), scaled by 2^13.
    std::vector<float> win((size_t)W);
    make_window(c.window, W, win.data());
    const double two_pi = 6.283185307179586476925286766559;
    p.dfrag.assign((size_t)KS * 8 * 64 * 8, 0);
    for (int ks = 0; ks < KS; ks++)
        for (int m = 0; m < 4; m++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    const int r = 16 * (m & 1) + (l & 15), n = 8 * (ks + KS * (l >> 4)) + j;   // k-step ks, lane group g: sample block ks + KS g
                    const bool imag = m >= 2;
                    double v = 0.0;
                    if (r < F && n < W) {
                        const int kn = (int)(((int64_t)(g.f0 + r) * n) % N);      // exact angle reduction
                        const double ang = two_pi * (double)kn / (double)N;
                        v = (double)win[(size_t)n] * (imag ? -std::sin(ang) : std::cos(ang)) * 8192.0;
                        if (imag && g.f0 + r == 0) v = 0.0;                       // DC is real (:323 drops the packed Nyquist)
                    }
                    uint16_t hi, lo;
                    split_half(v, hi, lo);
                    p.dfrag[((((size_t)ks * 4 + m) * 2 + 0) * 64 + l) * 8 + j] = hi;
                    p.dfrag[((((size_t)ks * 4 + m) * 2 + 1) * 64 + l) * 8 + j] = lo;
                }


    // ---- the symmetric-fold kernel (kernels_fused_s.hip): geometry, LDS layout per wave, folded basis.
    // Applicable when the window is symmetric about its centre, w[n] == w[W - n] (every window of WindowType.createWindow,
    // CircularShortTimeFourierTransform.swift:19-28, is: checked on the table itself), W is 64, 128, 192 or 256, the gap keeps
    // frames 16-byte aligned, and a tile's samples leave room in the wave's share of the LDS.
    {
        d.s_ok = 0;
        d.s_padp = 0;
        d.s_cs8 = 0;
        d.s_cs8_rc = 0;
        bool sym = (W % 64 == 0) && (g.gap % 4 == 0);
        // (to an ulp of the fp32 table: the two sides are averaged below, which moves a coefficient by 2^-25 of itself at most)
        for (int nn = 1; nn < W && sym; nn++) sym = std::fabs((double)win[(size_t)nn] - (double)win[(size_t)(W - nn)]) <= 2.4e-7 * std::fabs((double)win[(size_t)nn]) + 1e-30;
        const int TP = (T + 1) / 2 * 2;                          // taps stored per frame row (even: the row stride then spreads 16 rows over the banks)
        const int HQ = (H + 3) / 4;                              // quads of hidden units: one -> 8 waves a workgroup, more -> 4 (kernels_fused_s.hip)
        const int PSs = 4 * HQ * TP + 4;
        const int row_bytes = ((T - 1 + kFusedSTileFrames) * PSs + 8 * HQ + 128) * 4;   // rows, the zero quad and the dump quad, lane-group constants
        const int span = (kFusedSTileFrames - 1) * hop + W;
        // hops that are multiples of 64 floats put every frame of a tile on the same LDS banks: under a 256-sample window the ring
        // is then padded by a quad after every `padp` floats (the power of two that divides the hop; one DMA instruction per
        // padded piece of a chunk).  Instantiated for one quad of units (two would spill a register or three).  Not for
        // 128-sample windows: there the pieces' DMA instructions cost more than the conflicts (1.65 against 1.47 ms at hop 64,
        // profiles/r03_padded_ring_ab.txt)
        int padp = 0;
        if (hop % 64 == 0 && W == 256 && HQ == 1) padp = (hop % 128 == 0) ? 128 : 64;
        if (padp > W / 2) padp = 0;
#ifdef SYLDET_PLAN_NOPAD                 // (diagnostic builds, tools/knockouts.sh fused_plan.cpp ...: never the shipped library)
        padp = 0;
#endif
        int chunk_bytes = 1024 + (padp ? 16 * (256 / padp) : 0);
        auto ring_chunks = [&](int waves) { return std::min(30, (160 * 1024 / waves - (row_bytes + 15) / 16 * 16) / chunk_bytes - 1); };
        auto fits = [&](int rc) { return rc >= (span + 255) / 256 + 1 && kFusedSTileFrames * hop <= rc * 256; };
        // two waves a SIMD where the rows of tap products leave room for the ring in a 20 KB share of the LDS (one or two quads of units)
        // (one quad: always -- it is instantiated for 8 waves only)
        if (padp && !fits(ring_chunks(kFusedSBlock / 64))) {     // (the padding costs a chunk: hop 192 under a 256-sample window keeps the plain ring)
            padp = 0;
            chunk_bytes = 1024;
        }
        const int s_waves = (!wide_band && (HQ == 1 || (HQ == 2 && fits(ring_chunks(kFusedSBlock / 64))))) ? kFusedSBlock / 64 : kFusedSBlock / 128;
        const int per_wave = 160 * 1024 / s_waves;
        const int RC = ring_chunks(s_waves);
        if (sym && H <= 16 && c.n_layers == 2 && fits(RC)) {
            d.s_ok = 1;
            d.s_waves = s_waves;
            d.s_padp = (padp && s_waves == kFusedSBlock / 64) ? padp : 0;
            // (the slot permutation spreads the frames over the banks when a frame's start moves on by an odd number of quads --
            // the padding counts)
            d.s_perm = (hop / 4 + (d.s_padp ? hop / d.s_padp : 0)) % 2 == 1 ? 1 : 0;
#ifdef SYLDET_PLAN_NOPERM
            d.s_perm = (hop / 4) % 2 == 1 ? 1 : 0;
#endif
            d.s_ring_chunks = RC; d.s_pstride = PSs; d.s_tp = TP;
            d.s_lds_wave = per_wave;
            // hop 128 (where the padded ring is in use): the same share of the LDS as slots of 1280 bytes, no mirror chunk -- the
            // twice-folded form's other ring layout (kernels_fused_s.hip, CS8); a tile's span and the next tile's eight chunks
            d.s_cs8 = 0;
            d.s_cs8_rc = 0;
            if (d.s_padp == 128 && hop == 128) {
                const int rc8 = std::min(16, (per_wave - (row_bytes + 15) / 16 * 16) / 1280);
                if (rc8 >= (span + 255) / 256 + 1) {
                    d.s_cs8 = 1;
                    d.s_cs8_rc = rc8;
                }
            }
            const int K2 = W / 64, c0 = W / 2;
            // A operands, lane l: row l & 15 of its tile, k = 8 (l >> 4) + j -> folded position m = 32 ks + k.
            // s rows (real part): w[c0 + m] cos(2 pi k m / N), half of it at m = 0 (s[0] = 2 x[c0]);
            // d rows (imaginary part): -w[c0 + m] sin(2 pi k m / N); slot m = 0 takes the frame's first sample: +w[0] sin(2 pi k (W/2) / N).
            p.sfrag.assign((size_t)K2 * 2 * 2 * 2 * 64 * 8, 0);
            for (int ks = 0; ks < K2; ks++)
                for (int which = 0; which < 2; which++)
                    for (int m2 = 0; m2 < 2; m2++)
                        for (int l = 0; l < 64; l++)
                            for (int j = 0; j < 8; j++) {
                                const int r = 16 * m2 + (l & 15), m = 32 * ks + 8 * (l >> 4) + j;
                                double v = 0.0;
                                if (r < F) {
                                    const int k = g.f0 + r;
                                    const int kn = (int)(((int64_t)k * m) % N);
                                    const double ang = two_pi * (double)kn / (double)N;
                                    const double wm = 0.5 * ((double)win[(size_t)(c0 + m)] + (double)win[(size_t)(c0 - m)]);
                                    if (which == 0) {
                                        v = wm * std::cos(ang) * (m == 0 ? 0.5 : 1.0);
                                    } else if (m == 0) {
                                        const int kh = (int)(((int64_t)k * (W / 2)) % N);
                                        v = (double)win[0] * std::sin(two_pi * (double)kh / (double)N);
                                    } else {
                                        v = -wm * std::sin(ang);
                                    }
                                    if (which == 1 && k == 0) v = 0.0;            // DC is real (:323 drops the packed Nyquist)
                                    v *= 8192.0;
                                }
                                uint16_t hi, lo;
                                split_half(v, hi, lo);
                                const size_t base = (((((size_t)ks * 2 + which) * 2 + m2) * 2) * 64 + l) * 8 + j;
                                p.sfrag[base] = hi;
                                p.sfrag[base + 64 * 8] = lo;
                            }
            // the first sample's real part for the bins a lane holds in a result: 4 g + i (i < 4), 16 + 4 g + i - 4
            p.slone.assign(64 * 8, 0.0f);
            for (int l = 0; l < 64; l++)
                for (int i = 0; i < 8; i++) {
                    const int r = i < 4 ? 4 * (l >> 4) + i : 16 + 4 * (l >> 4) + (i - 4);
                    if (r < F) {
                        const int kh = (int)(((int64_t)(g.f0 + r) * (W / 2)) % N);
                        p.slone[(size_t)l * 8 + i] = (float)((double)win[0] * std::cos(two_pi * (double)kh / (double)N) * 8192.0);
                    }
                }
        }
    }

    // ---- the fold kernel's second fold (kernels_fused_s.hip, F2): for W == N the once-folded positions pair up again,
    // m with N/2 - m:  cos(theta_k (N/2 - m)) = (-1)^k cos(theta_k m),  sin(theta_k (N/2 - m)) = -(-1)^k sin(theta_k m)  (theta_k = 2 pi k / N).
    // With a[m] = w[c+m] (x[c+m] + x[c-m]), b[m] = w[c+m] (x[c+m] - x[c-m]) (the window now applied by the lanes, in fp32):
    //     Re' X[k] = sum_{m=1}^{N/4-1} cos(theta_k m) (a[m] + (-1)^k a[N/2-m])  +  [w[c] x[c] + (-1)^k w[0] x[0]]  +  a[N/4] cos(pi k / 2)
    //     Im' X[k] = -sum_{m=1}^{N/4-1} sin(theta_k m) (b[m] - (-1)^k b[N/2-m])  -  b[N/4] sin(pi k / 2)
    // Even and odd bins become GEMMs of their own with K = N/4 = 64 and ONE row tile each (a band of at most 32 bins has at most
    // 16 of either parity): 4 GEMMs x 2 k-steps x 3 products = 24 matrix instructions per 16 frames instead of 48, a basis of
    // 64 registers.  Slot 0 of the real rows (cos 0 = 1) takes the centre sample and the frame's first sample; slot 0 of the odd
    // imaginary rows (sin 0 = 0: free) takes b[N/4] against -sin(pi k / 2); a[N/4] meets the even real rows on the vector side.
    d.s2_ok = 0;
    d.s2_nt = 1;
    if (d.s_ok && W == 256 && N == 256 && H <= 16 && c.n_layers == 2) {
        d.s2_ok = 1;
        const int NT = wide_band ? 2 : 1;                 // row tiles per parity
        d.s2_nt = NT;
        const int ke0 = g.f0 + (g.f0 & 1), ko0 = g.f0 + 1 - (g.f0 & 1);          // first even / odd bin of the band
        d.s2_pe = ke0 - g.f0;
        d.s2_po = ko0 - g.f0;
        p.sfrag2.assign((size_t)2 * 4 * NT * 2 * 64 * 8, 0);
        for (int ks = 0; ks < 2; ks++)
            for (int gm = 0; gm < 4; gm++)
              for (int tau = 0; tau < NT; tau++)
                for (int l = 0; l < 64; l++)
                    for (int j = 0; j < 8; j++) {
                        const int r = 16 * tau + (l & 15), m = 32 * ks + 8 * (l >> 4) + j;
                        const int k = ((gm & 1) ? ko0 : ke0) + 2 * r;
                        double v = 0.0;
                        if (k < g.f1) {
                            const double ang = two_pi * (double)(((int64_t)k * m) % N) / (double)N;
                            if (gm < 2) v = std::cos(ang);                                     // (m = 0: 1, against the two lone samples)
                            else if (m > 0) v = -std::sin(ang);
                            else v = gm == 3 ? -std::sin(two_pi * (double)(((int64_t)k * (N / 4)) % N) / (double)N) : 0.0;   // b[N/4]'s slot
                            if (gm >= 2 && k == 0) v = 0.0;                                    // DC is real (:323)
                            v *= 8192.0;
                        }
                        uint16_t hi, lo;
                        split_half(v, hi, lo);
                        const size_t base = (((((size_t)ks * 4 + gm) * NT + tau) * 2) * 64 + l) * 8 + j;
                        p.sfrag2[base] = hi;
                        p.sfrag2[base + 64 * 8] = lo;
                    }
        // window coefficients of the positions lane group gq folds: w[c + m] and w[N/2 - (c - ...)] = w[m], the two sides of the
        // symmetric table averaged as above
        p.swin2.assign((size_t)4 * 16 * 2, 0.0f);
        for (int gq = 0; gq < 4; gq++)
            for (int ks = 0; ks < 2; ks++)
                for (int i = 0; i < 8; i++) {
                    const int m = 32 * ks + 8 * gq + i;
                    const double w1 = m == 0 ? (double)win[128] : 0.5 * ((double)win[(size_t)(128 + m)] + (double)win[(size_t)(128 - m)]);
                    const double w2 = m == 0 ? (double)win[0] : 0.5 * ((double)win[(size_t)m] + (double)win[(size_t)(256 - m)]);
                    p.swin2[((size_t)gq * 16 + ks * 8 + i) * 2 + 0] = (float)w1;
                    p.swin2[((size_t)gq * 16 + ks * 8 + i) * 2 + 1] = (float)w2;
                }
        p.s2c.assign((size_t)64 * 16, 0.0f);                // per lane: cos(pi k / 2) 2^13 of its even rows 4 tau + i, then w[192] at [8]
        for (int l = 0; l < 64; l++) {
            for (int tau = 0; tau < NT; tau++)
                for (int i = 0; i < 4; i++) {
                    const int k = ke0 + 2 * (16 * tau + 4 * (l >> 4) + i);
                    if (k < g.f1) p.s2c[(size_t)l * 16 + 4 * tau + i] = (float)(std::cos(two_pi * (double)(((int64_t)k * (N / 4)) % N) / (double)N) * 8192.0);
                }
            p.s2c[(size_t)l * 16 + 8] = (float)(0.5 * ((double)win[192] + (double)win[64]));
        }
    }

    // |X[k]| <= (sum_n |D[k][n]|) * max|x|: with samples scaled below 2^14 the column shift keeps |X| * 2^(cse-shift) < 2^13
    {
        double rowsum_max = 1.0;
        for (int r = 0; r < F; r++) {
            double sr = 0.0, si = 0.0;
            for (int n = 0; n < W; n++) {
                const int kn = (int)(((int64_t)(g.f0 + r) * n) % N);
                const double ang = two_pi * (double)kn / (double)N;
                sr += std::fabs((double)win[(size_t)n] * std::cos(ang));
                si += std::fabs((double)win[(size_t)n] * std::sin(ang));
            }
            rowsum_max = std::max(rowsum_max, std::sqrt(sr * sr + si * si));
        }
        d.col_shift = (int)std::ceil(std::log2(rowsum_max)) + 1;   // 2^14 * rowsum * 2^-shift <= 2^13
    }

    // ---- folded first layer, one fragment pair per tap: A operand of v_mfma_f32_16x16x32_f16, lane l holds
    // row l&15 (hidden unit), k = 8*(l>>4) + j (bin):  W'_t[h][f] = W0[h][t*F+f] * a[t*F+f], scaled by 2^wexp.
    double wmax = 0.0;
    for (int h = 0; h < H; h++)
        for (int i = 0; i < I; i++) wmax = std::max(wmax, std::fabs((double)L0.weights[(size_t)h * I + i] * a[(size_t)i]));
    int wexp = 0;
    if (wmax > 0.0) wexp = 13 - (int)std::ceil(std::log2(wmax));
    wexp = std::max(-100, std::min(100, wexp));
    const double wscale = std::ldexp(1.0, wexp);
    d.w_unscale = (float)std::ldexp(1.0, -wexp);
    p.afrag.assign((size_t)T * 2 * 64 * 8, 0);
    for (int t = 0; t < T; t++)
        for (int l = 0; l < 64; l++)
            for (int j = 0; j < 8; j++) {
                const int h = l & 15, bin = 8 * (l >> 4) + j;
                double v = 0.0;
                if (h < H && bin < F) v = (double)L0.weights[(size_t)h * I + t * F + bin] * a[(size_t)(t * F + bin)] * wscale;
                uint16_t hi, lo;
                split_half(v, hi, lo);
                p.afrag[(((size_t)t * 2 + 0) * 64 + l) * 8 + j] = hi;
                p.afrag[(((size_t)t * 2 + 1) * 64 + l) * 8 + j] = lo;
            }
    // the same layer with all taps as the rows of one A operand (kernels_fused_r.hip: every column meets every tap once)
    p.afrag_t.assign((size_t)3 * 2 * 64 * 8, 0);
    if (H <= 4)
        for (int m = 0; m < 3; m++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    const int r = 16 * m + (l & 15), t = r / 4, h = r % 4, gq = l >> 4, bin = j < 4 ? 4 * gq + j : 16 + 4 * gq + (j - 4);
                    double v = 0.0;
                    if (t < T && h < H && bin < F) v = (double)L0.weights[(size_t)h * I + t * F + bin] * a[(size_t)(t * F + bin)] * wscale;
                    uint16_t hi, lo;
                    split_half(v, hi, lo);
                    p.afrag_t[(((size_t)m * 2 + 0) * 64 + l) * 8 + j] = hi;
                    p.afrag_t[(((size_t)m * 2 + 1) * 64 + l) * 8 + j] = lo;
                }
    // ... in the bin order of the twice-folded result: k = 8 g + j -> band bin 8 g + pe + 2 j (j < 4), 8 g + po + 2 (j - 4) (j >= 4)
    p.afrag_t2.assign((size_t)3 * d.s2_nt * 2 * 64 * 8, 0);
    if (H <= 4 && d.s2_ok)
        for (int m = 0; m < 3; m++)
          for (int tau = 0; tau < d.s2_nt; tau++)                 // (one k-step of the tap GEMM per 32 bins)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    const int r = 16 * m + (l & 15), t = r / 4, h = r % 4, gq = l >> 4;
                    const int bin = 32 * tau + (j < 4 ? 8 * gq + d.s2_pe + 2 * j : 8 * gq + d.s2_po + 2 * (j - 4));
                    double v = 0.0;
                    if (t < T && h < H && bin < F) v = (double)L0.weights[(size_t)h * I + t * F + bin] * a[(size_t)(t * F + bin)] * wscale;
                    uint16_t hi, lo;
                    split_half(v, hi, lo);
                    p.afrag_t2[((((size_t)m * d.s2_nt + tau) * 2 + 0) * 64 + l) * 8 + j] = hi;
                    p.afrag_t2[((((size_t)m * d.s2_nt + tau) * 2 + 1) * 64 + l) * 8 + j] = lo;
                }
    // ... the wider layers' fragments (afrag_w below) in that bin order
    {
        const int HQ = (H + 3) / 4;
        p.afrag_w2.assign((size_t)3 * HQ * 2 * 64 * 8, 0);
        if (HQ > 1 && d.s2_ok && d.s2_nt == 1)
            for (int m = 0; m < 3; m++)
                for (int q = 0; q < HQ; q++)
                    for (int l = 0; l < 64; l++)
                        for (int j = 0; j < 8; j++) {
                            const int r = l & 15, t = 4 * m + r / 4, h = 4 * q + r % 4, gq = l >> 4;
                            const int bin = j < 4 ? 8 * gq + d.s2_pe + 2 * j : 8 * gq + d.s2_po + 2 * (j - 4);
                            double v = 0.0;
                            if (t < T && h < H && bin < F) v = (double)L0.weights[(size_t)h * I + t * F + bin] * a[(size_t)(t * F + bin)] * wscale;
                            uint16_t hi, lo;
                            split_half(v, hi, lo);
                            p.afrag_w2[((((size_t)m * HQ + q) * 2 + 0) * 64 + l) * 8 + j] = hi;
                            p.afrag_w2[((((size_t)m * HQ + q) * 2 + 1) * 64 + l) * 8 + j] = lo;
                        }
    }
    // ... and for 5 .. 16 hidden units (kernels_fused_s.hip, HQ = ceil(H / 4) quads): row tile (m, q), row 4 g + i of it = tap
    // 4 m + g, unit 4 q + i -- lane group g of a result then holds tap 4 m + g for every quad, as with one quad
    {
        const int HQ = (H + 3) / 4;
        p.afrag_w.assign((size_t)3 * HQ * 2 * 64 * 8, 0);
        if (HQ > 1)
            for (int m = 0; m < 3; m++)
                for (int q = 0; q < HQ; q++)
                    for (int l = 0; l < 64; l++)
                        for (int j = 0; j < 8; j++) {
                            const int r = l & 15, t = 4 * m + r / 4, h = 4 * q + r % 4, gq = l >> 4, bin = j < 4 ? 4 * gq + j : 16 + 4 * gq + (j - 4);
                            double v = 0.0;
                            if (t < T && h < H && bin < F) v = (double)L0.weights[(size_t)h * I + t * F + bin] * a[(size_t)(t * F + bin)] * wscale;
                            uint16_t hi, lo;
                            split_half(v, hi, lo);
                            p.afrag_w[((((size_t)m * HQ + q) * 2 + 0) * 64 + l) * 8 + j] = hi;
                            p.afrag_w[((((size_t)m * HQ + q) * 2 + 1) * 64 + l) * 8 + j] = lo;
                        }
    }
    p.bias0.resize((size_t)H);
    p.rvec.resize((size_t)H);
    for (int h = 0; h < H; h++) {
        double sb = (double)L0.biases[h], sr = 0.0;
        for (int i = 0; i < I; i++) {
            const double w = (double)L0.weights[(size_t)h * I + i];
            sb += w * b[(size_t)i];
            sr += w * a[(size_t)i];
        }
        p.bias0[(size_t)h] = (float)sb;
        p.rvec[(size_t)h] = (float)sr;
    }
    if (c.n_layers == 2) {
        const syldet_layer_t &L1 = c.layers[1];
        p.w1.assign(L1.weights, L1.weights + (size_t)L1.inputs * (size_t)L1.outputs);
        p.b1.assign(L1.biases, L1.biases + L1.outputs);
    }
    for (int k = 0; k < c.n_output_fns; k++) {
        const syldet_fn_t &f = c.output_fns[k];
        p.out_params.push_back(f.y);
        p.out_params.insert(p.out_params.end(), f.gains, f.gains + n_out);
        p.out_params.insert(p.out_params.end(), f.x_offsets, f.x_offsets + n_out);
    }
    // ---- precision guard (kernels.hpp, FixItem).  Sensitivity of an output to the network's input vector u, as a bound:
    // |dz_h| <= ||W'_h||_2 ||du||_2 (W' = W0 o a, the folded first layer in true units), transfer slopes <= 1 (LogSig 1/4),
    // |dy_o| <= sum_h |w1[o][h]| |dz_h|, every reverse output map divides by |gain_o|.
    {
        auto slope = [](int tf) { return tf == SYLDET_TF_LOGSIG ? 0.25 : 1.0; };
        std::vector<double> rown((size_t)H, 0.0);
        for (int h = 0; h < H; h++) {
            double s = 0.0;
            for (int i = 0; i < I; i++) {
                const double w = (double)L0.weights[(size_t)h * I + i] * a[(size_t)i];
                s += w * w;
            }
            rown[(size_t)h] = std::sqrt(s) * slope(L0.transfer);
        }
        double lip = 0.0;
        for (int o = 0; o < n_out; o++) {
            double lo = 0.0;
            if (c.n_layers == 2) {
                for (int h = 0; h < H; h++) lo += std::fabs((double)c.layers[1].weights[(size_t)o * H + h]) * rown[(size_t)h];
                lo *= slope(c.layers[1].transfer);
            } else {
                lo = rown[(size_t)o];
            }
            for (int k = 0; k < c.n_output_fns; k++) lo /= std::max(1e-30, std::fabs((double)c.output_fns[k].gains[o]));
            lip = std::max(lip, lo);
        }
        if (!(lip > 1e-30)) lip = 1e-30;
        if (!(lip < 1e30)) lip = 1e30;
        // The same chain as a root-sum-square (errors of different inputs and units carry random signs): what the arithmetic's
        // OWN relative error does to an output.  hi/lo-split operands and three products leave about 2^-21.4 (3.6e-7) of a frame's
        // column level in every bin (measured against the fp64 anchor: DESIGN 7).  Behind a normaliser that is 3.6e-7 of a
        // unit-norm vector whatever the recording's level; WITHOUT one the network sees the columns at the recording's level and
        // the same relative error grows with it -- a window of norm U moves output o by about 3.6e-7 U / sqrt(I) g_o,
        // g_o^2 = sum_h (w1[o][h] ||W'_h||_2)^2.  An fp32 FFT is eight times closer there, so once that expectation passes a
        // quarter of the 1e-5 bar the evaluation is recomputed exactly (fixup_kernel) rather than reported at a level-dependent bar.
        double grss = 0.0;
        for (int o = 0; o < n_out; o++) {
            double g2 = 0.0;
            if (c.n_layers == 2) {
                for (int h = 0; h < H; h++) {
                    const double t = (double)c.layers[1].weights[(size_t)o * H + h] * rown[(size_t)h];
                    g2 += t * t;
                }
                g2 *= slope(c.layers[1].transfer) * slope(c.layers[1].transfer);
            } else {
                g2 = rown[(size_t)o] * rown[(size_t)o];
            }
            double go = std::sqrt(g2);
            for (int k = 0; k < c.n_output_fns; k++) go /= std::max(1e-30, std::fabs((double)c.output_fns[k].gains[o]));
            grss = std::max(grss, go);
        }
        {
            const double k = 3.6e-7 * grss / (std::sqrt((double)I) * 2.5e-6);
            d.guard_loud = (float)std::min(std::max(k * k, 1e-30), 1e30);
        }
        // Grid floors, in units of a stored column value.  The 8-wave kernel splits every column into f16 hi + lo at the
        // pass's scale: half an f16 subnormal step (2^-25) per bin plus the sample grid's share.  The register-resident-basis
        // kernel gives every frame its own column exponent, which leaves the sample grid: 2^-25 per sample of a pass scaled to
        // 2^14, through the basis (bound 2^-26 per bin, typically 2^-31).
        const double eps_out = 2e-6, phi_c = std::ldexp(1.0, -24), phi_r = std::ldexp(1.0, -27), rel = std::ldexp(1.0, -21);
        const double sqI = std::sqrt((double)I);
        auto sq = [](double v) { return (float)std::min(v * v, 1e30); };
        d.guard_r = sq(lip * sqI * phi_r / eps_out);
        d.guard_c = sq(lip * sqI * phi_c / eps_out);
        d.guard_c_range = (float)std::min((norm == 2 ? 4.0 : 2.0) * lip * sqI * phi_c / eps_out, 1e30);
        d.guard_range_r = (float)std::min((norm == 2 ? 4.0 : 2.0) * lip * sqI * phi_r / eps_out, 1e30);
        // no normaliser: every column has to stand on its own (the reference's error is relative to the frame, and nothing
        // divides a quiet column's error by a loud neighbour's norm): the smallest column sum of squares of the window
        d.guard_rel_r = sq(std::sqrt((double)F) * phi_r / rel);
        d.guard_rel_c = sq(std::sqrt((double)F) * phi_c / rel);
        // no normaliser: the floor in true units is phi 2^(col_shift - se); it matters once  lip sqrt(I) phi 2^(col_shift - se) > eps
        auto se_abs = [&](double phi) {
            const double v = std::log2(lip * sqI * phi / eps_out) + (double)d.col_shift;
            return (int)std::max(-200.0, std::min(200.0, std::ceil(v)));
        };
        d.guard_se_abs_r = se_abs(phi_r);
        d.guard_se_abs_c = se_abs(phi_c);
        // the spectrogram instantiation stores fp32 columns straight from the accumulators: the sample grid's floor alone, held
        // against 1e-6 of the column's largest value (its norm / sqrt(F) at least) or 1e-6 absolute, whichever is larger
        d.guard_spect = sq(std::sqrt((double)F) * phi_r / 1e-6);
        d.guard_se_abs_s = (int)std::max(-200.0, std::min(200.0, std::ceil(std::log2(phi_r / 1e-6) + (double)d.col_shift)));
    }
    if (!d.classic_ok && !fused_r_applicable(d) && !fused_s_applicable(d)) return no(classic_hop_ok ? "LDS budget exceeded" : "hop too large for the staging registers / the sample ring");
    p.koff.resize((size_t)KS * 4);
    for (int ks = 0; ks < KS; ks++)
        for (int h = 0; h < 4; h++) {
            const int o = 8 * (ks + KS * h);
            p.koff[(size_t)ks * 4 + h] = o + skew * (o / hop);
        }
    p.ok = true;
    return true;
}

bool make_mlpx_plan(const syldet_config_t &c, const syldet_geometry_t &g, MlpxPlan &p)
{
    p = MlpxPlan();
    auto no = [&p](const char *why) { p.reason = why; return false; };
    const int F = g.bins, T = c.time_range, I = g.inputs;
    // (log / dB columns are fine here: the chain starts with l2normalize, so what the f16 hi + lo split loses -- 2^-22 of the
    // frame's column norm -- is lost relative to the vector the network sees)
    if (c.spectrum == SYLDET_SPECTRUM_MAGNITUDE) return no("not |X| columns");
    if (c.n_layers != 2 || g.outputs != 1) return no("not two layers with one output");
    const syldet_layer_t &L0 = c.layers[0], &L1 = c.layers[1];
    const int H = L0.outputs;
    if (H > 4 || L0.transfer != SYLDET_TF_TANSIG || L1.transfer != SYLDET_TF_PURELIN) return no("not <= 4 TanSig units and a linear output");
    if (c.n_input_fns < 1 || c.input_fns[0].kind != SYLDET_FN_L2NORMALIZE) return no("input chain does not start with l2normalize");
    if (c.n_output_fns > 1) return no("more than one output map");
    if (F > 128) return no("more than 128 bins");
    // affine tail of the input chain:  x = a o v' + b  (MapMinMax.apply NeuralNet.swift:127-131, MapStd.apply :162-169)
    std::vector<double> a((size_t)I, 1.0), b((size_t)I, 0.0);
    for (int k = 1; k < c.n_input_fns; k++) {
        const syldet_fn_t &f = c.input_fns[k];
        if (f.kind != SYLDET_FN_MAPMINMAX && f.kind != SYLDET_FN_MAPSTD) return no("a normaliser follows another input function");
        for (int i = 0; i < I; i++) {
            a[(size_t)i] = a[(size_t)i] * (double)f.gains[i];
            b[(size_t)i] = (b[(size_t)i] - (double)f.x_offsets[i]) * (double)f.gains[i] + (double)f.y;
        }
    }
    if (T > 12) return no("timeRange above 12");
    // (1024-point frames: four blocks whatever the band, which is what the one-launch kernel of kernels_fft1k.hip is
    // instantiated for -- a narrow band then pays a few idle MFMAs instead of a second launch and the columns' trip through HBM)
    const int KB = (c.fourier_length == 1024 && c.window_length == 1024 && F % 4 == 0) ? 4 : (F <= 32 ? 1 : (F <= 64 ? 2 : 4));
    MlpxDesc &d = p.desc;
    d.F = F; d.T = T; d.KB = KB; d.H = H; d.rule = c.rule; d.scaling = c.scaling;
    d.col_stride = 32 * KB + 8;                      // 16-byte aligned rows that spread 16 consecutive rows over all banks
    d.p_stride = 48 + 4;                             // floats per frame of tap products: 12 taps x 4 units, padded likewise
    int off = 0;
    auto take = [&off](int bytes) { const int o = off; off += (bytes + 15) / 16 * 16; return o; };
    d.lds_afrag = take(3 * KB * 2 * 1024);
    d.lds_colh = take(kMlpxTile * d.col_stride * 2);
    d.lds_coll = take(kMlpxTile * d.col_stride * 2);
    d.lds_p = take(kMlpxTile * d.p_stride * 4);
    d.lds_pq = take(kMlpxTile * 32 * 4);             // sums of squares per quad of column values (at most 32 quads a frame)
    d.lds_ss = take(2 * kMlpxTile * 4);              // per-frame sums of squares, two tiles (parity)
    d.lds_red = take(2 * kMlpxTile * 4);             // per-frame scales 2^fe and their inverses
    d.lds_total = off;
    if (off > 160 * 1024) return no("LDS budget exceeded");
    // folded first layer (see make_fused_plan): W'_t[h][f] = W0[h][t*F+f] * a[t*F+f], scaled by 2^wexp, f16 hi + lo.
    // ALL taps are rows of one GEMM: row 4t + h of the A operand (three 16-row tiles hold 12 taps x 4 units), K = bins.
    // A operand of v_mfma_f32_16x16x32_f16: lane l holds row 16m + (l&15), k = 8*(l>>4) + j (bin 32 kb + k).
    double wmax = 0.0;
    for (int h = 0; h < H; h++)
        for (int i = 0; i < I; i++) wmax = std::max(wmax, std::fabs((double)L0.weights[(size_t)h * I + i] * a[(size_t)i]));
    int wexp = wmax > 0.0 ? 13 - (int)std::ceil(std::log2(wmax)) : 0;
    wexp = std::max(-100, std::min(100, wexp));
    const double wscale = std::ldexp(1.0, wexp);
    d.w_unscale = (float)std::ldexp(1.0, -wexp);
    p.afrag.assign((size_t)3 * KB * 2 * 64 * 8, 0);
    for (int m = 0; m < 3; m++)
        for (int kb = 0; kb < KB; kb++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    const int r = 16 * m + (l & 15), t = r / 4, h = r % 4, bin = 32 * kb + 8 * (l >> 4) + j;
                    double v = 0.0;
                    if (t < T && h < H && bin < F) v = (double)L0.weights[(size_t)h * I + t * F + bin] * a[(size_t)(t * F + bin)] * wscale;
                    uint16_t hi, lo;
                    split_half(v, hi, lo);
                    p.afrag[((((size_t)m * KB + kb) * 2 + 0) * 64 + l) * 8 + j] = hi;
                    p.afrag[((((size_t)m * KB + kb) * 2 + 1) * 64 + l) * 8 + j] = lo;
                }
    p.bias0.assign(4, 0.0f);
    p.w1.assign(4, 0.0f);
    for (int h = 0; h < H; h++) {
        double sb = (double)L0.biases[h];
        for (int i = 0; i < I; i++) sb += (double)L0.weights[(size_t)h * I + i] * b[(size_t)i];
        p.bias0[(size_t)h] = (float)sb;
        p.w1[(size_t)h] = L1.weights[h];
    }
    d.b1 = L1.biases[0];
    d.oa = 0.0f; d.og = 1.0f; d.ob = 0.0f;           // reverse map (y - y0) / gain + xoff (NeuralNet.swift:137-142 / :175-180)
    if (c.n_output_fns == 1) {
        if (c.output_fns[0].kind != SYLDET_FN_MAPMINMAX && c.output_fns[0].kind != SYLDET_FN_MAPSTD) return no("output function is not a map");
        d.oa = c.output_fns[0].y; d.og = c.output_fns[0].gains[0]; d.ob = c.output_fns[0].x_offsets[0];
    }
    p.ok = true;
    return true;
}

bool make_bdft_plan(const syldet_config_t &c, const syldet_geometry_t &g, const MlpxPlan &mx, BdftPlan &p)
{
    p = BdftPlan();
    const int N = c.fourier_length, W = c.window_length, hop = g.hop, F = g.bins, T = c.time_range, I = g.inputs;
    // the matrix-core network stage's class with 128-bin columns, |X| columns (linear, ln or dB of them)
    // (its columns are 128 bins wide whatever the band: the owner gives it a copy of the stage's descriptor with those strides)
    if (!mx.ok || c.spectrum == SYLDET_SPECTRUM_MAGNITUDE) return false;
    // frames of four or two whole blocks (75 % or 50 % overlap), no gap, no zero padding; 2 or 4 k-steps of 32 folded positions a
    // block; 512 points and up (a frame costs this kernel the same whatever its length, the FFT kernels less the shorter it is:
    // 256-point frames at hop 256 take 1.89 ms here against 1.36 as two launches, 512-point ones at hop 256 2.02 against 2.66)
    if (W != N || g.gap != 0 || N < 512 || (hop != 128 && hop != 256) || (4 * hop != N && 2 * hop != N)) return false;
    // the window as a short cosine sum (WindowType.createWindow, CircularShortTimeFourierTransform.swift:19-28; Blackman's five taps
    // along the bins are not built)
    double a0, a1;
    if (c.window == SYLDET_WINDOW_HAMMING) { a0 = 0.54; a1 = -0.46; }
    else if (c.window == SYLDET_WINDOW_HANNING) { a0 = 0.5; a1 = -0.5; }
    else if (c.window == SYLDET_WINDOW_NONE) { a0 = 1.0; a1 = 0.0; }
    else return false;
    if (g.f0 < 1 || T > 12) return false;
    // the bin under the band is a neighbour of its first one; 128 bins from a multiple of 4 (where the spectrum has no more, from 0)
    const int kb0 = std::min((g.f0 - 1) / 4 * 4, std::max(0, N / 2 - 128));
    if (g.f0 + F + 1 > kb0 + 128 || kb0 + 128 > N / 2) return false;
    const int KS = hop / 64, c0 = hop / 2;
    const double two_pi = 6.283185307179586476925286766559, theta = two_pi * (double)c0 / (double)N;
    BdftDesc &d = p.desc;
    d.hop = hop; d.kb0 = kb0; d.f0 = g.f0; d.R = N / hop;
    d.a0 = (float)a0; d.a1c = (float)(0.5 * a1 * std::cos(theta)); d.a1s = (float)(0.5 * a1 * std::sin(theta));
    // A operands, lane l: row l & 15 of its tile (bin kb0 + 16 w + row), k = 8 (l >> 4) + j -> folded position m = 32 ks + k.
    // cosine rows: cos(2 pi k m / N), half of it at m = 0 (s[0] = 2 x[c]); sine rows: -sin(2 pi k m / N), slot m = 0 takes the
    // block's first sample: +sin(theta k).  Scaled by 2^13.
    p.basis.assign((size_t)8 * 2 * KS * 2 * 64 * 8, 0);
    for (int w = 0; w < 8; w++)
        for (int which = 0; which < 2; which++)
            for (int ks = 0; ks < KS; ks++)
                for (int l = 0; l < 64; l++)
                    for (int j = 0; j < 8; j++) {
                        const int k = kb0 + 16 * w + (l & 15), m = 32 * ks + 8 * (l >> 4) + j;
                        const int km = (int)(((int64_t)k * m) % N), kc = (int)(((int64_t)k * c0) % N);
                        const double ang = two_pi * (double)km / (double)N;
                        double v;
                        if (which == 0) v = std::cos(ang) * (m == 0 ? 0.5 : 1.0);
                        else v = m == 0 ? std::sin(two_pi * (double)kc / (double)N) : -std::sin(ang);
                        v *= 8192.0;
                        uint16_t hi, lo;
                        split_half(v, hi, lo);
                        const size_t base = (((((size_t)w * 2 + which) * KS + ks) * 2) * 64 + l) * 8 + j;
                        p.basis[base] = hi;
                        p.basis[base + 64 * 8] = lo;
                    }
    p.cre.assign((size_t)8 * 64 * 4, 0.0f);
    for (int w = 0; w < 8; w++)
        for (int l = 0; l < 64; l++)
            for (int i = 0; i < 4; i++) {
                const int k = kb0 + 16 * w + 4 * (l >> 4) + i, kc = (int)(((int64_t)k * c0) % N);
                p.cre[((size_t)w * 64 + l) * 4 + i] = (float)(std::cos(two_pi * (double)kc / (double)N) * 8192.0);
            }
    // the first layer with all taps as rows (make_mlpx_plan's table) with K = bin - kb0: zero weights outside the band
    std::vector<double> a((size_t)I, 1.0);
    for (int k = 1; k < c.n_input_fns; k++)
        for (int i = 0; i < I; i++) a[(size_t)i] *= (double)c.input_fns[k].gains[i];
    const syldet_layer_t &L0 = c.layers[0];
    const int H = L0.outputs;
    const double wscale = 1.0 / (double)mx.desc.w_unscale;
    p.afrag.assign((size_t)3 * 4 * 2 * 64 * 8, 0);
    for (int m = 0; m < 3; m++)
        for (int kb = 0; kb < 4; kb++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    const int r = 16 * m + (l & 15), t = r / 4, h = r % 4, bin = kb0 + 32 * kb + 8 * (l >> 4) + j - g.f0;
                    double v = 0.0;
                    if (t < T && h < H && bin >= 0 && bin < F) v = (double)L0.weights[(size_t)h * I + t * F + bin] * a[(size_t)(t * F + bin)] * wscale;
                    uint16_t hi, lo;
                    split_half(v, hi, lo);
                    p.afrag[((((size_t)m * 4 + kb) * 2 + 0) * 64 + l) * 8 + j] = hi;
                    p.afrag[((((size_t)m * 4 + kb) * 2 + 1) * 64 + l) * 8 + j] = lo;
                }
    p.ok = true;
    return true;
}

void fused_segmentation(FusedDesc &d, int64_t E, int C)
{
    // A workgroup segment = `runs` consecutive passes of one channel, emitting pass * runs - (T-1) evaluations.  Both kernels
    // run one workgroup per CU, and a workgroup pays a prologue and a drain worth about two passes: segments as long as the
    // batch allows (up to kMax passes; longer measured the same), their number per channel rounded up so that the grid is
    // whole rounds of the 256 CUs where the channel count allows (64 channels x 16 segments = 4 rounds for the benchmark
    // batch on the 64-frame-pass kernel); small batches get more, shorter segments to fill the chip.
    const int64_t frames = E + d.T - 1;
    auto segment = [&](int pass, int kMax, int &runs_out, int &seg_out) {
        const int64_t max_evals = (int64_t)kMax * pass - (d.T - 1);
        int64_t segs = std::max<int64_t>(1, (E + max_evals - 1) / max_evals);
        if ((int64_t)C * segs >= 1024) {
            int g = C, m = 256;
            while (g) { const int t = m % g; m = g; g = t; }   // m = gcd(C, 256)
            const int64_t mult = 256 / m;
            segs = (segs + mult - 1) / mult * mult;
        } else {
            segs = std::max<int64_t>(segs, std::min<int64_t>((1024 + C - 1) / C, (frames + pass - 1) / pass));
        }
        const int64_t per = (E + segs - 1) / segs;       // evaluations per segment
        int64_t rr = (per + (d.T - 1) + pass - 1) / pass;
        rr = std::max<int64_t>(1, std::min<int64_t>(kMax, rr));
        runs_out = (int)rr;
        seg_out = (int)(rr * pass - (d.T - 1));
    };
    segment(kFusedTileFrames, 64, d.runs, d.seg_evals);
    segment(kFusedRTileFrames, 128, d.r_runs, d.r_seg_evals);
    // The symmetric-fold kernel: a segment belongs to a WAVE (eight to a workgroup, a workgroup per CU).  A wave pays its
    // prologue (the basis from L2, the first tile's samples) once and T - 1 frames of lead-in, so segments are as long as the
    // batch allows: their number is the smallest multiple of 2048 / gcd(C, 2048) per channel that keeps every wave slot of the
    // 256 CUs busy in whole rounds (64 channels: 32 segments of ~249 tiles for the benchmark batch); short batches get what
    // fills the chip.
    {
#ifndef SYLDET_S_MAXTILES                // (diagnostic builds: shorter wave segments = more rounds of workgroups; measured no faster, MEASUREMENTS R3.3)
#define SYLDET_S_MAXTILES 512
#endif
        const int64_t max_tiles = SYLDET_S_MAXTILES;
        const int64_t max_evals = max_tiles * kFusedSTileFrames - (d.T - 1);
        int64_t segs = std::max<int64_t>(1, (E + max_evals - 1) / max_evals);
        const int sw = d.s_waves > 0 ? d.s_waves : kFusedSBlock / 64;
        const int64_t slots = 256 * sw;
        if ((int64_t)C * segs >= slots) {
            int gg = C, m = (int)slots;
            while (gg) { const int t = m % gg; m = gg; gg = t; }
            const int64_t mult = slots / m;
            segs = (segs + mult - 1) / mult * mult;
        } else {
            segs = std::max<int64_t>(segs, std::min<int64_t>((slots + C - 1) / C, (frames + 4 * kFusedSTileFrames - 1) / (4 * kFusedSTileFrames)));
        }
        segs = (segs + sw - 1) / sw * sw;                        // whole workgroups per channel
        int64_t per = (E + segs - 1) / segs;
        int64_t tl = (per + (d.T - 1) + kFusedSTileFrames - 1) / kFusedSTileFrames;
        tl = std::max<int64_t>(1, tl);
        d.s_seg_evals = (int)(tl * kFusedSTileFrames - (d.T - 1));
    }
}

}  // namespace sd
