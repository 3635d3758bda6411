/*
 * syldet_oracle.c -- CPU restatement of the reference hot path (see syldet_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (no reference goldens exist; see header).
 *
 * Two arithmetic modes:
 *   ORC_F32  the "port": fp32, same operation order as the Swift/vDSP call sequence
 *            (window multiply, even/odd packing, N/2 radix-2 complex FFT, real split,
 *            x2 scaling, zvabs, /2, sequential fp32 reductions).  Built with
 *            -ffp-contract=off so no FMA contraction changes the rounding sequence.
 *   ORC_F64  the anchor: same data (fp32 samples, fp32 window table, fp32 weights) but
 *            every operation in double, the spectrum by the DFT definition.
 * Reference paths are cited relative to the reference root.
 */
#include "syldet_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ geometry */

/* CircularShortTimeFourierTransform.init, Common/CircularShortTimeFourierTransform.swift:61-96
 * (gap/overlap split :66-73, overlap < W :76, pow2 :82, W <= N :86) and
 * SyllableDetector.init, Common/SyllableDetector.swift:46-60 (shape checks).         */
int orc_geometry(const orc_config_t *c, orc_geom_t *g)
{
    int N = c->fourier_length, W = c->window_length, ov = c->window_overlap;
    if (N <= 0 || (N & (N - 1)) != 0) return -1;
    if (W <= 0 || W > N) return -2;
    if (ov >= W) return -3;
    g->gap = ov < 0 ? -ov : 0;
    g->overlap = ov < 0 ? 0 : ov;
    g->hop = g->gap + W - g->overlap;
    int f0, f1;
    if (orc_frequency_index_range(N, c->sampling_rate, c->freq_lo, c->freq_hi, &f0, &f1) != 0) return -4;
    g->f0 = f0; g->f1 = f1; g->F = f1 - f0;
    g->I = g->F * c->time_range;
    if (c->n_layers < 1 || c->n_layers > ORC_MAX_LAYERS) return -5;
    for (int l = 1; l < c->n_layers; l++)
        if (c->layers[l - 1].outputs != c->layers[l].inputs) return -6;   /* NeuralNet.swift:248-254 */
    if (c->layers[0].inputs != g->I) return -7;                            /* SyllableDetector.swift:52-55 */
    g->n_out = c->layers[c->n_layers - 1].outputs;
    if (c->n_thresholds != g->n_out) return -8;                            /* SyllableDetector.swift:58-60 */
    if (c->time_range < 1) return -9;
    return 0;
}

/* frequencyIndexRange, Common/CircularShortTimeFourierTransform.swift:166-191 */
int orc_frequency_index_range(int N, double fs, double lo, double hi, int *f0, int *f1)
{
    if (!(lo >= 0.0 && hi > lo)) return -1;
    int half = N / 2;
    double from_frequency = (double)N / fs;
    int start = (int)ceil(from_frequency * lo);
    if (start >= half) return -1;
    int end = (int)floor(from_frequency * hi) + 1;
    if (end < start) return -1;
    if (end > half) end = half;
    *f0 = start; *f1 = end;
    return 0;
}

/* WindowType.createWindow, Common/CircularShortTimeFourierTransform.swift:19-28.
 * vDSP_hamm_window / vDSP_hann_window / vDSP_blkman_window with flag 0: full-length,
 * periodic (denominator N), HANN "denormalised" (0.5 factor).  The table is float.  */
void orc_window(int type, int len, float *w)
{
    for (int n = 0; n < len; n++) {
        double a = 2.0 * M_PI * (double)n / (double)len, v;
        switch (type) {
        case ORC_WIN_HAMMING:  v = 0.54 - 0.46 * cos(a); break;
        case ORC_WIN_HANNING:  v = 0.5 * (1.0 - cos(a)); break;
        case ORC_WIN_BLACKMAN: v = 0.42 - 0.5 * cos(a) + 0.08 * cos(2.0 * a); break;
        default:               v = 1.0; break;
        }
        w[n] = (float)v;
    }
}

/* extractPower availability rule :286-288 and consume :299-302 => frame j covers
 * samples [j*hop+gap, j*hop+gap+W), exists iff j*hop + gap + W <= S.                */
int64_t orc_count_frames(const orc_config_t *c, int64_t S)
{
    orc_geom_t g;
    if (orc_geometry(c, &g) != 0) return -1;
    int64_t need = (int64_t)g.gap + c->window_length;
    if (S < need) return 0;
    return (S - need) / g.hop + 1;
}

/* processNewValue needs T columns, consumes one (SyllableDetector.swift:164-178)    */
int64_t orc_count_evals(const orc_config_t *c, int64_t S)
{
    int64_t J = orc_count_frames(c, S);
    if (J < 0) return J;
    return J >= c->time_range ? J - c->time_range + 1 : 0;
}

/* ------------------------------------------------------------------ STFT frame */

static int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

/* Precomputed state of one CircularShortTimeFourierTransform instance: the window
 * table (:103-104), the FFT setup's twiddle table (vDSP_create_fftsetup :100), the
 * zeroed pad buffer (:107-110) and the split-complex scratch (:116-122).            */
typedef struct {
    int N, W, M, L;
    float *win;
    int *rev;
    float *twr, *twi;   /* stage twiddles, concatenated: for len=2,4,..,M: len/2 entries */
    float *swr, *swi;   /* real-split twiddles e^{-2 pi i k / N}, k < M                  */
    float *xw, *zr, *zi;
    double *tc, *ts;    /* fp64 anchor tables cos/sin(2 pi m / N)                         */
} stft_plan_t;

static void plan_init(stft_plan_t *p, const orc_config_t *c)
{
    int N = c->fourier_length, W = c->window_length, M = N / 2;
    p->N = N; p->W = W; p->M = M; p->L = ilog2(M);
    p->win = (float *)malloc(sizeof(float) * (size_t)W);
    orc_window(c->window, W, p->win);
    p->rev = (int *)malloc(sizeof(int) * (size_t)(M > 0 ? M : 1));
    for (int m = 0; m < M; m++) {
        int r = 0;
        for (int b = 0; b < p->L; b++) if (m & (1 << b)) r |= 1 << (p->L - 1 - b);
        p->rev[m] = r;
    }
    p->twr = (float *)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1));
    p->twi = (float *)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1));
    int off = 0;
    for (int len = 2; len <= M; len <<= 1) {
        for (int k = 0; k < len / 2; k++) {
            double a = -2.0 * M_PI * (double)k / (double)len;
            p->twr[off + k] = (float)cos(a); p->twi[off + k] = (float)sin(a);
        }
        off += len / 2;
    }
    p->swr = (float *)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1));
    p->swi = (float *)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1));
    for (int k = 0; k < M; k++) {
        double a = -2.0 * M_PI * (double)k / (double)N;
        p->swr[k] = (float)cos(a); p->swi[k] = (float)sin(a);
    }
    p->xw = (float *)calloc((size_t)N, sizeof(float));              /* vDSP_vclr :110: the tail W..N stays 0 */
    p->zr = (float *)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1));
    p->zi = (float *)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1));
    p->tc = (double *)malloc(sizeof(double) * (size_t)N);
    p->ts = (double *)malloc(sizeof(double) * (size_t)N);
    for (int m = 0; m < N; m++) {
        p->tc[m] = cos(2.0 * M_PI * (double)m / (double)N);
        p->ts[m] = sin(2.0 * M_PI * (double)m / (double)N);
    }
}

static void plan_free(stft_plan_t *p)
{
    free(p->win); free(p->rev); free(p->twr); free(p->twi); free(p->swr); free(p->swi);
    free(p->xw); free(p->zr); free(p->zi); free(p->tc); free(p->ts);
}

/* fp32 port of extractPower/extractMagnitude,
 * Common/CircularShortTimeFourierTransform.swift:280-337 / :221-278.
 * Only bins [k0,k1) of the final magnitude pass are produced (the reference computes
 * all N/2 and the detector slices [f0,f1) afterwards, SyllableDetector.swift:140-148;
 * the values are identical).                                                         */
static void stft_frame_f32(const orc_config_t *c, stft_plan_t *p, const float *x, int k0, int k1, float *out)
{
    int N = p->N, W = p->W, M = p->M;
    float *xw = p->xw, *zr = p->zr, *zi = p->zi;
    (void)N;
    for (int n = 0; n < W; n++) xw[n] = x[n] * p->win[n];           /* vDSP_vmul :311 */
    /* vDSP_ctoz stride 2 :314-316: even samples -> realp, odd -> imagp (stored
     * bit-reversed here for an in-place decimation-in-time radix-2 transform)         */
    for (int m = 0; m < M; m++) { int r = p->rev[m]; zr[r] = xw[2 * m]; zi[r] = xw[2 * m + 1]; }
    /* vDSP_fft_zript radix 2 :320: complex FFT of length N/2 ...                       */
    int off = 0;
    for (int len = 2; len <= M; len <<= 1) {
        int half = len / 2;
        for (int s0 = 0; s0 < M; s0 += len) {
            for (int k = 0; k < half; k++) {
                float wr = p->twr[off + k], wi = p->twi[off + k];
                int s = s0 + k, t = s + half;
                float tr = zr[t] * wr - zi[t] * wi;
                float ti = zr[t] * wi + zi[t] * wr;
                zr[t] = zr[s] - tr; zi[t] = zi[s] - ti;
                zr[s] = zr[s] + tr; zi[s] = zi[s] + ti;
            }
        }
        off += half;
    }
    /* ... followed by the real split; vDSP's packed result is 2 x DFT, with 2X[0] in
     * realp[0] and 2X[N/2] in imagp[0].  imagp[0] is then cleared :323.               */
    float scale = c->power_mode ? 4.0f : 2.0f;                      /* :272 / :331 */
    for (int k = k0; k < k1; k++) {
        float re2, im2;
        if (k == 0) {
            re2 = 2.0f * (zr[0] + zi[0]);
            im2 = 0.0f;
        } else {
            float wr = p->swr[k], wi = p->swi[k];
            float ar = zr[k] + zr[M - k], ai = zi[k] - zi[M - k];   /* Z[k] + conj Z[M-k] */
            float br = zr[k] - zr[M - k], bi = zi[k] + zi[M - k];   /* Z[k] - conj Z[M-k] */
            float tr = br * wr - bi * wi, ti = br * wi + bi * wr;   /* w^k * B            */
            re2 = ar + ti;                                          /* 2X[k] = A - i w^k B */
            im2 = ai - tr;
        }
        float v = c->power_mode ? (re2 * re2 + im2 * im2)           /* vDSP_zvmags :270 */
                                : sqrtf(re2 * re2 + im2 * im2);     /* vDSP_zvabs  :329 */
        out[k - k0] = v / scale;                                    /* vDSP_vsdiv :273/:332 */
    }
}

/* fp64 anchor: |X[k]| (or |X[k]|^2) by the DFT definition over bins [k0,k1).          */
static void stft_frame_f64(const orc_config_t *c, const stft_plan_t *p, const float *x, int k0, int k1, double *out)
{
    int N = p->N, W = p->W;
    for (int k = k0; k < k1; k++) {
        double re = 0.0, im = 0.0;
        for (int n = 0; n < W; n++) {
            double v = (double)x[n] * (double)p->win[n];
            int m = (int)(((int64_t)k * n) % N);
            re += v * p->tc[m];
            im -= v * p->ts[m];
        }
        if (k == 0) im = 0.0;   /* DC is real; Nyquist was discarded :323 */
        out[k - k0] = c->power_mode ? (re * re + im * im) : sqrt(re * re + im * im);
    }
}

int orc_stft_frame(const orc_config_t *c, const float *x, int precision, double *out)
{
    int M = c->fourier_length / 2;
    stft_plan_t p;
    plan_init(&p, c);
    if (precision == ORC_F32) {
        float *o = (float *)malloc(sizeof(float) * (size_t)M);
        stft_frame_f32(c, &p, x, 0, M, o);
        for (int k = 0; k < M; k++) out[k] = (double)o[k];
        free(o);
    } else {
        stft_frame_f64(c, &p, x, 0, M, out);
    }
    plan_free(&p);
    return 0;
}

/* processFourierData, Common/SyllableDetector.swift:134-151: every available frame,
 * sliced to [f0,f1), appended column after column.                                   */
int64_t orc_spectrogram(const orc_config_t *c, const float *samples, int64_t S, int precision, double *cols)
{
    orc_geom_t g;
    if (orc_geometry(c, &g) != 0) return -1;
    int64_t J = orc_count_frames(c, S);
    stft_plan_t p;
    plan_init(&p, c);
    if (precision == ORC_F32) {
        float *o = (float *)malloc(sizeof(float) * (size_t)g.F);
        for (int64_t j = 0; j < J; j++) {
            stft_frame_f32(c, &p, samples + j * g.hop + g.gap, g.f0, g.f1, o);
            for (int f = 0; f < g.F; f++) cols[j * g.F + f] = (double)o[f];
        }
        free(o);
    } else {
        for (int64_t j = 0; j < J; j++)
            stft_frame_f64(c, &p, samples + j * g.hop + g.gap, g.f0, g.f1, cols + j * g.F);
    }
    plan_free(&p);
    return J;
}

/* ------------------------------------------------------------------ neural net */

/* Input processing functions, Common/NeuralNet.swift:41-182, fp32 op order.          */
static void in_fn_f32(const orc_fn_t *fn, float *v, int n)
{
    switch (fn->kind) {
    case ORC_FN_L2NORMALIZE: {                                  /* :47-59 */
        float sumsq = 0.0f;
        for (int i = 0; i < n; i++) sumsq += v[i] * v[i];       /* vDSP_svesq */
        float s = sqrtf(sumsq);
        for (int i = 0; i < n; i++) v[i] = v[i] / s;            /* vDSP_vsdiv */
        break;
    }
    case ORC_FN_NORMALIZE: {                                    /* :69-96 */
        float mn = v[0], mx = v[0];
        for (int i = 1; i < n; i++) { if (v[i] < mn) mn = v[i]; if (v[i] > mx) mx = v[i]; }
        float range = mx - mn;
        if (range == 0.0f) { for (int i = 0; i < n; i++) v[i] = -1.0f; break; }
        float slope = 2.0f / range, intercept = (0.0f - mn - mx) / range;
        for (int i = 0; i < n; i++) v[i] = v[i] * slope + intercept;   /* vDSP_vsmsa */
        break;
    }
    case ORC_FN_NORMALIZESTD: {                                 /* :105-108, vDSP_normalize: population sigma */
        float sum = 0.0f;
        for (int i = 0; i < n; i++) sum += v[i];
        float mean = sum / (float)n, ss = 0.0f;
        for (int i = 0; i < n; i++) { float d = v[i] - mean; ss += d * d; }
        float sd = sqrtf(ss / (float)n);
        for (int i = 0; i < n; i++) v[i] = (v[i] - mean) / sd;
        break;
    }
    case ORC_FN_MAPMINMAX:                                      /* :127-131 vDSP_vsbm then vDSP_vsadd */
        for (int i = 0; i < n; i++) { float t = (v[i] - fn->xoff[i]) * fn->gain[i]; v[i] = t + fn->y; }
        break;
    case ORC_FN_MAPSTD:                                         /* :162-169 */
        for (int i = 0; i < n; i++) {
            float t = (v[i] - fn->xoff[i]) * fn->gain[i];
            v[i] = (fn->y != 0.0f) ? t + fn->y : t;
        }
        break;
    default: break;
    }
}

static void in_fn_f64(const orc_fn_t *fn, double *v, int n)
{
    switch (fn->kind) {
    case ORC_FN_L2NORMALIZE: {
        double sumsq = 0.0;
        for (int i = 0; i < n; i++) sumsq += v[i] * v[i];
        double s = sqrt(sumsq);
        for (int i = 0; i < n; i++) v[i] = v[i] / s;
        break;
    }
    case ORC_FN_NORMALIZE: {
        double mn = v[0], mx = v[0];
        for (int i = 1; i < n; i++) { if (v[i] < mn) mn = v[i]; if (v[i] > mx) mx = v[i]; }
        double range = mx - mn;
        if (range == 0.0) { for (int i = 0; i < n; i++) v[i] = -1.0; break; }
        double slope = 2.0 / range, intercept = (0.0 - mn - mx) / range;
        for (int i = 0; i < n; i++) v[i] = v[i] * slope + intercept;
        break;
    }
    case ORC_FN_NORMALIZESTD: {
        double sum = 0.0;
        for (int i = 0; i < n; i++) sum += v[i];
        double mean = sum / (double)n, ss = 0.0;
        for (int i = 0; i < n; i++) { double d = v[i] - mean; ss += d * d; }
        double sd = sqrt(ss / (double)n);
        for (int i = 0; i < n; i++) v[i] = (v[i] - mean) / sd;
        break;
    }
    case ORC_FN_MAPMINMAX:
        for (int i = 0; i < n; i++) v[i] = (v[i] - (double)fn->xoff[i]) * (double)fn->gain[i] + (double)fn->y;
        break;
    case ORC_FN_MAPSTD:
        for (int i = 0; i < n; i++) v[i] = (v[i] - (double)fn->xoff[i]) * (double)fn->gain[i] + (double)fn->y;
        break;
    default: break;
    }
}

/* Transfer functions, Common/NeuralNet.swift:185-228 */
static float transfer_f32(int tf, float x)
{
    switch (tf) {
    case ORC_TF_TANSIG: return tanhf(x);                        /* vvtanhf :192 */
    case ORC_TF_LOGSIG: { float e = expf(-1.0f * x); e = e + 1.0f; return 1.0f / e; }  /* :199-213 */
    case ORC_TF_SATLIN: return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);              /* vDSP_vclip :226 */
    default: return x;
    }
}
static double transfer_f64(int tf, double x)
{
    switch (tf) {
    case ORC_TF_TANSIG: return tanh(x);
    case ORC_TF_LOGSIG: return 1.0 / (exp(-x) + 1.0);
    case ORC_TF_SATLIN: return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);
    default: return x;
    }
}

/* Output processing (reverse maps), Common/NeuralNet.swift:133-143, :171-181:
 * vsadd(-y) ; vDSP_vdiv(gains, dest) = dest / gains ; vadd(xoff)                     */
static void out_fn_f32(const orc_fn_t *fn, float *v, int n)
{
    if (fn->kind != ORC_FN_MAPMINMAX && fn->kind != ORC_FN_MAPSTD) return;
    float neg = 0.0f - fn->y;
    for (int i = 0; i < n; i++) { float t = v[i] + neg; t = t / fn->gain[i]; v[i] = t + fn->xoff[i]; }
}
static void out_fn_f64(const orc_fn_t *fn, double *v, int n)
{
    if (fn->kind != ORC_FN_MAPMINMAX && fn->kind != ORC_FN_MAPSTD) return;
    for (int i = 0; i < n; i++) v[i] = (v[i] - (double)fn->y) / (double)fn->gain[i] + (double)fn->xoff[i];
}

static int max_width(const orc_config_t *c)
{
    int m = c->layers[0].inputs;
    for (int l = 0; l < c->n_layers; l++) if (c->layers[l].outputs > m) m = c->layers[l].outputs;
    return m;
}

/* NeuralNet.apply, Common/NeuralNet.swift:294-326; NeuralNetLayer.apply :366-377     */
static void net_apply_f32(const orc_config_t *c, const float *in, float *out, float *bufA, float *bufB)
{
    int n = c->layers[0].inputs;
    memcpy(bufA, in, sizeof(float) * (size_t)n);                 /* applyAndCopy / PassThrough :261-272 */
    for (int k = 0; k < c->n_in_fns; k++) in_fn_f32(&c->in_fns[k], bufA, n);
    float *cur = bufA, *nxt = bufB;
    for (int l = 0; l < c->n_layers; l++) {
        const orc_layer_t *L = &c->layers[l];
        for (int o = 0; o < L->outputs; o++) {
            float acc = 0.0f;                                    /* vDSP_mmul M=outputs N=1 P=inputs :368 */
            const float *wrow = L->weights + (size_t)o * (size_t)L->inputs;
            for (int i = 0; i < L->inputs; i++) acc += wrow[i] * cur[i];
            acc = acc + L->biases[o];                            /* vDSP_vadd :371 */
            nxt[o] = transfer_f32(L->transfer, acc);             /* :374 */
        }
        float *t = cur; cur = nxt; nxt = t;
    }
    int no = c->layers[c->n_layers - 1].outputs;
    memcpy(out, cur, sizeof(float) * (size_t)no);
    for (int k = 0; k < c->n_out_fns; k++) out_fn_f32(&c->out_fns[k], out, no);
}

static void net_apply_f64(const orc_config_t *c, const double *in, double *out, double *bufA, double *bufB)
{
    int n = c->layers[0].inputs;
    memcpy(bufA, in, sizeof(double) * (size_t)n);
    for (int k = 0; k < c->n_in_fns; k++) in_fn_f64(&c->in_fns[k], bufA, n);
    double *cur = bufA, *nxt = bufB;
    for (int l = 0; l < c->n_layers; l++) {
        const orc_layer_t *L = &c->layers[l];
        for (int o = 0; o < L->outputs; o++) {
            double acc = 0.0;
            const float *wrow = L->weights + (size_t)o * (size_t)L->inputs;
            for (int i = 0; i < L->inputs; i++) acc += (double)wrow[i] * cur[i];
            acc += (double)L->biases[o];
            nxt[o] = transfer_f64(L->transfer, acc);
        }
        double *t = cur; cur = nxt; nxt = t;
    }
    int no = c->layers[c->n_layers - 1].outputs;
    memcpy(out, cur, sizeof(double) * (size_t)no);
    for (int k = 0; k < c->n_out_fns; k++) out_fn_f64(&c->out_fns[k], out, no);
}

int orc_net_apply(const orc_config_t *c, const float *in, int precision, double *out)
{
    int mw = max_width(c), no = c->layers[c->n_layers - 1].outputs, n = c->layers[0].inputs;
    if (precision == ORC_F32) {
        float *a = (float *)malloc(sizeof(float) * (size_t)mw), *b = (float *)malloc(sizeof(float) * (size_t)mw);
        float *o = (float *)malloc(sizeof(float) * (size_t)no);
        net_apply_f32(c, in, o, a, b);
        for (int i = 0; i < no; i++) out[i] = (double)o[i];
        free(a); free(b); free(o);
    } else {
        double *a = (double *)malloc(sizeof(double) * (size_t)mw), *b = (double *)malloc(sizeof(double) * (size_t)mw);
        double *i64 = (double *)malloc(sizeof(double) * (size_t)n);
        for (int i = 0; i < n; i++) i64[i] = (double)in[i];
        net_apply_f64(c, i64, out, a, b);
        free(a); free(b); free(i64);
    }
    return 0;
}

/* spectrogram scaling, Common/SyllableDetector.swift:184-212 (intended math; the
 * reference frees the scaled buffer before use in the .db/.log branches :188-191)    */
static float scale_f32(int scaling, float v)
{
    if (scaling == ORC_SCALE_LOG) return logf(v);                       /* vvlogf :207 */
    if (scaling == ORC_SCALE_DB)  return 20.0f * log10f(v / 1.0f);      /* vDSP_vdbcon flag 1, ref 1 :195 */
    return v;
}
static double scale_f64(int scaling, double v)
{
    if (scaling == ORC_SCALE_LOG) return log(v);
    if (scaling == ORC_SCALE_DB)  return 20.0 * log10(v);
    return v;
}

/* detection rule: SyllableDetector.swift:27-31 (first) / TrackDetector.swift:72-77 (any):
 * Double(Float out) >= Double threshold                                              */
static uint8_t detect(const orc_config_t *c, const float *out, int n_out, int rule)
{
    if (rule == ORC_RULE_FIRST) return (double)out[0] >= c->thresholds[0];
    for (int i = 0; i < n_out; i++) if ((double)out[i] >= c->thresholds[i]) return 1;
    return 0;
}

/* processNewValue over a whole channel, Common/SyllableDetector.swift:153-217        */
int64_t orc_run(const orc_config_t *c, const float *samples, int64_t S, int precision, int rule,
                float *outputs, uint8_t *flags, double *outputs64)
{
    orc_geom_t g;
    if (orc_geometry(c, &g) != 0) return -1;
    int64_t J = orc_count_frames(c, S), E = orc_count_evals(c, S);
    if (E <= 0) return 0;
    double *cols = (double *)malloc(sizeof(double) * (size_t)(J * g.F));
    orc_spectrogram(c, samples, S, precision, cols);
    int mw = max_width(c);
    if (precision == ORC_F32) {
        float *v = (float *)malloc(sizeof(float) * (size_t)g.I);
        float *a = (float *)malloc(sizeof(float) * (size_t)mw), *b = (float *)malloc(sizeof(float) * (size_t)mw);
        for (int64_t e = 0; e < E; e++) {
            /* T consecutive columns, oldest first :180-181 */
            for (int i = 0; i < g.I; i++) v[i] = scale_f32(c->scaling, (float)cols[e * g.F + i]);
            net_apply_f32(c, v, outputs + e * g.n_out, a, b);
            if (outputs64) for (int i = 0; i < g.n_out; i++) outputs64[e * g.n_out + i] = (double)outputs[e * g.n_out + i];
            if (flags) flags[e] = detect(c, outputs + e * g.n_out, g.n_out, rule);
        }
        free(v); free(a); free(b);
    } else {
        double *v = (double *)malloc(sizeof(double) * (size_t)g.I), *o = (double *)malloc(sizeof(double) * (size_t)g.n_out);
        double *a = (double *)malloc(sizeof(double) * (size_t)mw), *b = (double *)malloc(sizeof(double) * (size_t)mw);
        for (int64_t e = 0; e < E; e++) {
            for (int i = 0; i < g.I; i++) v[i] = scale_f64(c->scaling, cols[e * g.F + i]);
            net_apply_f64(c, v, o, a, b);
            for (int i = 0; i < g.n_out; i++) {
                outputs[e * g.n_out + i] = (float)o[i];
                if (outputs64) outputs64[e * g.n_out + i] = o[i];
            }
            if (flags) flags[e] = detect(c, outputs + e * g.n_out, g.n_out, rule);
        }
        free(v); free(o); free(a); free(b);
    }
    free(cols);
    return E;
}

/* TrackDetector.init/process, SyllableDetectorCLI/TrackDetector.swift:39-43 (first
 * index), :67-68 (advance by W - ov per evaluation), :80,:99 (debounce).             */
int64_t orc_detections(const orc_config_t *c, const uint8_t *flags, int64_t E, double debounce_seconds,
                       int64_t *idx, int64_t cap)
{
    int W = c->window_length, ov = c->window_overlap;
    int64_t next_output = (int64_t)W + (int64_t)(W - ov) * (c->time_range - 1);
    if (ov < 0) next_output -= ov;
    int64_t debounce_frames = (int64_t)(debounce_seconds * c->sampling_rate);   /* :19-26, Int() truncates */
    int64_t debounce_until = -1, n = 0;
    for (int64_t e = 0; e < E; e++) {
        int64_t cur = next_output;
        next_output += W - ov;
        if (flags[e] && debounce_until < cur) {
            if (n < cap) idx[n] = cur;
            n++;
            debounce_until = cur + debounce_frames;
        }
    }
    return n;
}

/* ------------------------------------------------------------------ streaming */

/* TPCircularBuffer, Common/TPCircularBuffer/TPCircularBuffer.h:102-189 and .c:43-124:
 * byte ring, length rounded up to a page, reads never wrap (mirrored mapping).       */
typedef struct { uint8_t *buf; int32_t length, head, tail, fill; } ring_t;

static void ring_init(ring_t *r, int32_t length)
{
    int32_t page = 4096;
    r->length = (length + page - 1) / page * page;                /* round_page, .c:49 */
    r->buf = (uint8_t *)calloc(2, (size_t)r->length);              /* second half = the mirror */
    r->head = r->tail = r->fill = 0;
}
static int ring_produce_bytes(ring_t *r, const void *src, int32_t len)   /* .h:177-185 */
{
    if (r->length - r->fill < len) return 0;
    const uint8_t *s = (const uint8_t *)src;
    /* one memcpy into the ring (.h:181; the mirrored mapping makes a write past the end land at the start) + the mirror */
    int32_t first = r->length - r->head < len ? r->length - r->head : len;
    memcpy(r->buf + r->head, s, (size_t)first);
    memcpy(r->buf + r->head + r->length, s, (size_t)first);
    if (len > first) {
        memcpy(r->buf, s + first, (size_t)(len - first));
        memcpy(r->buf + r->length, s + first, (size_t)(len - first));
    }
    r->head = (r->head + len) % r->length;
    r->fill += len;
    return 1;
}
static const void *ring_tail(const ring_t *r, int32_t *avail)             /* .h:102-106 */
{
    *avail = r->fill;
    return r->fill ? r->buf + r->tail : NULL;
}
static void ring_consume(ring_t *r, int32_t amount)                       /* .h:116-120 */
{
    r->tail = (r->tail + amount) % r->length;
    r->fill -= amount;
}

struct orc_stream {
    orc_config_t cfg;
    orc_geom_t g;
    int precision;
    ring_t samples;   /* CircularShortTimeFourierTransform.buffer, 409600 B :61 */
    ring_t features;  /* SyllableDetector.buffer, F*T*512 bytes, SyllableDetector.swift:63-67 */
    stft_plan_t plan;
    float *last;
};

orc_stream_t *orc_stream_create(const orc_config_t *c, int precision)
{
    orc_geom_t g;
    if (orc_geometry(c, &g) != 0) return NULL;
    orc_stream_t *s = (orc_stream_t *)calloc(1, sizeof(*s));
    s->cfg = *c; s->g = g; s->precision = precision;
    ring_init(&s->samples, 409600);
    ring_init(&s->features, g.F * c->time_range * 512);
    plan_init(&s->plan, c);
    s->last = (float *)calloc((size_t)g.n_out, sizeof(float));       /* lastOutputs zeros :70 */
    return s;
}

void orc_stream_destroy(orc_stream_t *s)
{
    if (!s) return;
    free(s->samples.buf); free(s->features.buf); plan_free(&s->plan); free(s->last); free(s);
}

/* appendAudioData -> appendData, SyllableDetector.swift:129-132, CircularSTFT.swift:197-201 */
int orc_stream_append(orc_stream_t *s, const float *data, int64_t n)
{
    return ring_produce_bytes(&s->samples, data, (int32_t)(n * 4)) ? 0 : -1;
}

/* processFourierData, SyllableDetector.swift:134-151 + extractPower CircularSTFT.swift:280-337 */
static int stream_process_fourier(orc_stream_t *s)
{
    const orc_config_t *c = &s->cfg;
    int32_t avail;
    const float *tail = (const float *)ring_tail(&s->samples, &avail);
    if (avail < (s->g.gap + c->window_length) * 4) return 0;              /* :286-288 */
    const float *x = tail + s->g.gap;                                       /* :294-296 */
    int M = c->fourier_length / 2;
    float *col = (float *)malloc(sizeof(float) * (size_t)M);
    if (s->precision == ORC_F32) {
        stft_frame_f32(c, &s->plan, x, 0, M, col);
    } else {
        double *d = (double *)malloc(sizeof(double) * (size_t)M);
        stft_frame_f64(c, &s->plan, x, 0, M, d);
        for (int k = 0; k < M; k++) col[k] = (float)d[k];
        free(d);
    }
    ring_consume(&s->samples, (s->g.gap + c->window_length - s->g.overlap) * 4);   /* :299-302 */
    int ok = ring_produce_bytes(&s->features, col + s->g.f0, s->g.F * 4);           /* :143-148 */
    free(col);
    return ok ? 1 : -1;
}

int orc_stream_process_new_value(orc_stream_t *s)                          /* SyllableDetector.swift:153-217 */
{
    const orc_config_t *c = &s->cfg;
    int r;
    while ((r = stream_process_fourier(s)) == 1) {}
    if (r < 0) return -1;
    int32_t avail;
    const float *p = (const float *)ring_tail(&s->features, &avail);
    if (!p) return 0;
    if (avail < s->g.I * 4) return 0;
    float *v = (float *)malloc(sizeof(float) * (size_t)s->g.I);
    for (int i = 0; i < s->g.I; i++) v[i] = scale_f32(c->scaling, p[i]);
    int mw = max_width(c);
    if (s->precision == ORC_F32) {
        float *a = (float *)malloc(sizeof(float) * (size_t)mw), *b = (float *)malloc(sizeof(float) * (size_t)mw);
        net_apply_f32(c, v, s->last, a, b);
        free(a); free(b);
    } else {
        double *vd = (double *)malloc(sizeof(double) * (size_t)s->g.I), *o = (double *)malloc(sizeof(double) * (size_t)s->g.n_out);
        double *a = (double *)malloc(sizeof(double) * (size_t)mw), *b = (double *)malloc(sizeof(double) * (size_t)mw);
        for (int i = 0; i < s->g.I; i++) vd[i] = scale_f64(c->scaling, (double)p[i]);
        net_apply_f64(c, vd, o, a, b);
        for (int i = 0; i < s->g.n_out; i++) s->last[i] = (float)o[i];
        free(vd); free(o); free(a); free(b);
    }
    free(v);
    ring_consume(&s->features, s->g.F * 4);                                 /* defer :175-178 */
    return 1;
}

void orc_stream_last_outputs(const orc_stream_t *s, float *out)
{
    memcpy(out, s->last, sizeof(float) * (size_t)s->g.n_out);
}
int orc_stream_last_detected(const orc_stream_t *s)                        /* :27-31 */
{
    return (double)s->last[0] >= s->cfg.thresholds[0];
}
int orc_stream_seen_syllable(orc_stream_t *s)                              /* :220-230 */
{
    int ret = 0;
    while (orc_stream_process_new_value(s) == 1) if (orc_stream_last_detected(s)) ret = 1;
    return ret;
}

/* The reference's whole consumer loop for one channel, frame at a time (TrackDetector.process,
 * SyllableDetectorCLI/TrackDetector.swift:45-105: append a decoded sample buffer, then `while detector.processNewValue()`),
 * with `chunk` samples per buffer.  outputs [E][n_out] / flags [E] may be NULL (timing runs).  Returns E.               */
int64_t orc_stream_run(const orc_config_t *c, int precision, const float *samples, int64_t S, int64_t chunk,
                       float *outputs, uint8_t *flags)
{
    orc_stream_t *s = orc_stream_create(c, precision);
    if (!s) return -1;
    int64_t e = 0;
    for (int64_t pos = 0; pos < S; pos += chunk) {
        int64_t n = S - pos < chunk ? S - pos : chunk;
        if (orc_stream_append(s, samples + pos, n) != 0) { orc_stream_destroy(s); return -2; }
        while (orc_stream_process_new_value(s) == 1) {
            if (outputs) memcpy(outputs + e * s->g.n_out, s->last, sizeof(float) * (size_t)s->g.n_out);
            if (flags) flags[e] = (uint8_t)orc_stream_last_detected(s);
            e++;
        }
    }
    orc_stream_destroy(s);
    return e;
}

/* ------------------------------------------------------------------ resampler */

/* ResamplerLinear, Common/Resampler.swift:29-70.  vDSP_vramp: C[n] = A + n*B;
 * vDSP_vlint: C[n] = A[floor b] + (b - floor b) * (A[floor b + 1] - A[floor b]).
 * The carry `offset = indices[last] + step - Float(n_in - 1)` :65 is restated as
 * written (it is one sample larger than the distance to the next buffer's origin).    */
void orc_resampler_init(orc_resampler_t *r, double rate_in, double rate_out)
{
    r->step = (float)(rate_in / rate_out); r->last = 0.0f; r->offset = 0.0f;
}
int64_t orc_resampler_count(const orc_resampler_t *r, int64_t n_in)
{
    return (int64_t)(((float)n_in - r->offset) / r->step);                 /* :40 */
}
int64_t orc_resampler_run(orc_resampler_t *r, const float *data, int64_t n_in, float *out)
{
    int across = r->offset < 0.0f;                                          /* :37 */
    int64_t n_out = orc_resampler_count(r, n_in);
    if (n_out <= 0) return 0;
    float last_index = 0.0f;
    for (int64_t i = 0; i < n_out; i++) {
        float b = r->offset + (float)i * r->step;                           /* vDSP_vramp :52 */
        if (i == n_out - 1) last_index = b;
        if (i == 0 && across) b = 0.0f;                                     /* :54-56 */
        int64_t k = (int64_t)floorf(b);
        float frac = b - (float)k;
        float a0 = data[k], a1 = (k + 1 < n_in) ? data[k + 1] : data[k];    /* reference reads one past the end here */
        out[i] = a0 + frac * (a1 - a0);                                     /* vDSP_vlint :59 */
    }
    if (across) {
        out[0] = (r->last * (0.0f - r->offset)) + (data[0] * (1.0f + r->offset));   /* :61-63 */
        if (n_out == 1) last_index = 0.0f;                                  /* indices[0] was overwritten :55 */
    }
    r->offset = last_index + r->step - (float)(n_in - 1);                   /* :65 */
    r->last = data[n_in - 1];                                               /* :66 */
    return n_out;
}
