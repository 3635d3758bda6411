/*
 * syldet_oracle.h -- CPU restatement of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * there only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (gardner-lab/syllable-detector-swift) ships no
 * tests, golden vectors or audio fixtures, cannot be built here (Swift + Apple
 * Accelerate + Mach VM), and does its arithmetic inside closed-source vDSP/vForce.
 * This oracle restates the Swift call sequence with vDSP semantics taken from
 * Apple's public documentation; it is anchored by analytic known-answer tests and an
 * independent numpy cross-check (tests/test_oracle.py), not by reference outputs.
 *
 * Every function cites the reference file:line (relative to the reference root) it
 * follows.
 */
#ifndef SYLDET_ORACLE_H
#define SYLDET_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* WindowType, Common/CircularShortTimeFourierTransform.swift:12-29 */
enum { ORC_WIN_NONE = 0, ORC_WIN_HAMMING = 1, ORC_WIN_HANNING = 2, ORC_WIN_BLACKMAN = 3 };
/* SyllableDetectorConfig.Scaling, Common/SyllableDetectorConfig.swift:13-30 */
enum { ORC_SCALE_LINEAR = 0, ORC_SCALE_LOG = 1, ORC_SCALE_DB = 2 };
/* processing functions, Common/SyllableDetectorConfig.swift:128-168 */
enum { ORC_FN_L2NORMALIZE = 0, ORC_FN_NORMALIZE = 1, ORC_FN_NORMALIZESTD = 2,
       ORC_FN_MAPMINMAX = 3, ORC_FN_MAPSTD = 4 };
/* transfer functions, Common/NeuralNet.swift:185-228 */
enum { ORC_TF_TANSIG = 0, ORC_TF_LOGSIG = 1, ORC_TF_PURELIN = 2, ORC_TF_SATLIN = 3 };
/* arithmetic precision of a run */
enum { ORC_F32 = 32, ORC_F64 = 64 };
/* detection rule: live/GUI looks at output 0 only (SyllableDetector.swift:27-31),
 * the CLI at any output (TrackDetector.swift:72-77) */
enum { ORC_RULE_FIRST = 0, ORC_RULE_ANY = 1 };

#define ORC_MAX_FNS 8
#define ORC_MAX_LAYERS 8

typedef struct {
    int32_t kind;
    int32_t count;      /* length of xoff / gain (mapminmax, mapstd) */
    const float *xoff;
    const float *gain;
    float y;            /* yMin (mapminmax) or yMean (mapstd) */
} orc_fn_t;

typedef struct {
    int32_t inputs, outputs, transfer;
    const float *weights; /* row-major [outputs][inputs], NeuralNet.swift:368 */
    const float *biases;
} orc_layer_t;

typedef struct {
    double sampling_rate;
    int32_t fourier_length, window_length, window_overlap;
    double freq_lo, freq_hi;
    int32_t time_range;
    int32_t scaling;
    int32_t window;       /* the detector always uses ORC_WIN_HAMMING (SyllableDetector.swift:43) */
    int32_t power_mode;   /* 0: extractPower = |X| (detector); 1: extractMagnitude = |X|^2 */
    int32_t n_in_fns;
    orc_fn_t in_fns[ORC_MAX_FNS];
    int32_t n_layers;
    orc_layer_t layers[ORC_MAX_LAYERS];
    int32_t n_out_fns;
    orc_fn_t out_fns[ORC_MAX_FNS];
    int32_t n_thresholds;
    const double *thresholds;
} orc_config_t;

/* derived geometry */
typedef struct {
    int32_t gap, overlap, hop, f0, f1, F, I, n_out;
} orc_geom_t;

int orc_geometry(const orc_config_t *c, orc_geom_t *g);            /* 0 ok, <0 invalid */
void orc_window(int type, int len, float *w);
int orc_frequency_index_range(int N, double fs, double lo, double hi, int *f0, int *f1);
int64_t orc_count_frames(const orc_config_t *c, int64_t S);
int64_t orc_count_evals(const orc_config_t *c, int64_t S);

/* One STFT frame: x points at `window_length` samples; out gets N/2 values.      */
int orc_stft_frame(const orc_config_t *c, const float *x, int precision, double *out_halfspec);

/* Spectrogram columns of one channel, sliced to [f0,f1): cols is [J][F].           */
int64_t orc_spectrogram(const orc_config_t *c, const float *samples, int64_t S,
                        int precision, double *cols);

/* NeuralNet.apply on one length-I vector (input already scaled).                   */
int orc_net_apply(const orc_config_t *c, const float *in, int precision, double *out);

/* Whole channel, batch formulation: frames j at j*hop+gap, eval e over columns
 * e..e+T-1.  outputs [E][n_out] (float, the reference's lastOutputs), flags [E]
 * (rule applied), outputs64 optional [E][n_out] unrounded doubles.                 */
int64_t orc_run(const orc_config_t *c, const float *samples, int64_t S, int precision,
                int rule, float *outputs, uint8_t *flags, double *outputs64);

/* Detection sample indices + debounce, TrackDetector.swift:39-43,65-100.           */
int64_t orc_detections(const orc_config_t *c, const uint8_t *flags, int64_t E,
                       double debounce_seconds, int64_t *idx, int64_t cap);

/* Streaming restatement: two byte rings exactly as the reference drives them.      */
typedef struct orc_stream orc_stream_t;
orc_stream_t *orc_stream_create(const orc_config_t *c, int precision);
void orc_stream_destroy(orc_stream_t *s);
int orc_stream_append(orc_stream_t *s, const float *data, int64_t n);   /* <0: ring full */
int orc_stream_process_new_value(orc_stream_t *s);                      /* 1/0           */
void orc_stream_last_outputs(const orc_stream_t *s, float *out);
int orc_stream_last_detected(const orc_stream_t *s);
int orc_stream_seen_syllable(orc_stream_t *s);
/* the whole consumer loop of one channel, `chunk` samples per appended buffer (TrackDetector.swift:45-105); returns E */
int64_t orc_stream_run(const orc_config_t *c, int precision, const float *samples, int64_t S, int64_t chunk,
                       float *outputs, uint8_t *flags);

/* ResamplerLinear, Common/Resampler.swift:20-76 */
typedef struct { float step, last, offset; } orc_resampler_t;
void orc_resampler_init(orc_resampler_t *r, double rate_in, double rate_out);
int64_t orc_resampler_count(const orc_resampler_t *r, int64_t n_in);
int64_t orc_resampler_run(orc_resampler_t *r, const float *data, int64_t n_in, float *out);

#ifdef __cplusplus
}
#endif
#endif
