/*
 * syldet.h -- C ABI of libsyldet, the MI355X (gfx950) batched syllable-detection engine.
 *
 * This is the drop-in boundary for ONE path of gardner-lab/syllable-detector-swift:
 *   CircularShortTimeFourierTransform.extractPower  -> band slice -> sliding timeRange window
 *   -> NeuralNet.apply -> threshold,   as driven by SyllableDetector.processNewValue.
 *
 * The reference has no FFI for this path (the arithmetic is inline Swift calling Apple
 * Accelerate); its only C boundary is Common/Common-Bridging-Header.h:5, which exposes
 * TPCircularBuffer.h to Swift.  libsyldet is bound the same way (one more #include in
 * that bridging header), and every entry point below names the
 * reference interface (file:line, relative to the reference root) it stands in for.
 *
 * Conventions
 *   - plain C types only; the library copies every configuration array at create time;
 *     callers own all input/output buffers; the library owns its device memory;
 *   - return value: SYLDET_OK (0) or a negative syldet_status_t.  Where the reference
 *     calls fatalError (ring overflow, shape mismatch, bad FFT size) or throws
 *     ParseError, this ABI returns a status instead of aborting the host; data
 *     availability is reported as 1/0 like processNewValue's Bool;
 *   - `*_device` entry points take device pointers and a hipStream_t (passed as void*)
 *     and are asynchronous on that stream; the others take host pointers and block;
 *   - channels are independent detectors (Processor.swift:57-59: one SyllableDetector
 *     per channel; main.swift:86-89: one per track); batch layouts are channel-major;
 *   - threading: one producer (append) + one consumer (process/read) per channel, as
 *     TPCircularBuffer.h:14 guarantees in the reference (append never locks; it allocates
 *     once, on a channel's first samples); the host-pointer batch calls and the streaming
 *     consumers of one handle serialise on its staging buffers; the *_device batch calls are
 *     not re-entrant on one handle, and two of them must not be in flight at once on different
 *     streams either (a handle owns one scratch set and one work list of the precision guard:
 *     enqueue a handle's calls on one stream, or use a handle per stream); distinct handles are
 *     independent.
 *   - there is NO CPU fallback: without a gfx950 device create fails with
 *     SYLDET_ERR_NO_DEVICE.
 */
#ifndef SYLDET_H
#define SYLDET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYLDET_ABI_VERSION 1

typedef enum {
    SYLDET_OK = 0,
    SYLDET_ERR_INVALID_ARGUMENT   = -1,  /* NULL pointer, negative size, bad enum                        */
    SYLDET_ERR_FFT_SIZE           = -2,  /* CircularShortTimeFourierTransform.swift:82-88 fatalError     */
    SYLDET_ERR_OVERLAP            = -3,  /* CircularShortTimeFourierTransform.swift:76-78 fatalError     */
    SYLDET_ERR_FREQ_RANGE         = -4,  /* SyllableDetector.swift:46-48 fatalError                      */
    SYLDET_ERR_INPUT_MISMATCH     = -5,  /* SyllableDetector.swift:52-55 fatalError                      */
    SYLDET_ERR_THRESHOLD_MISMATCH = -6,  /* SyllableDetector.swift:58-60 fatalError                      */
    SYLDET_ERR_LAYER_SHAPE        = -7,  /* NeuralNet.swift:244-254, :341-349 fatalError                 */
    SYLDET_ERR_BUFFER_FULL        = -8,  /* CircularShortTimeFourierTransform.swift:199 "Insufficient space on buffer." */
    SYLDET_ERR_NO_DEVICE          = -9,  /* no gfx950 device / HIP runtime failure at create             */
    SYLDET_ERR_DEVICE             = -10, /* HIP error during a call (message via syldet_last_error)      */
    SYLDET_ERR_OUT_OF_MEMORY      = -11,
    SYLDET_ERR_PARSE_OPEN         = -20, /* ParseError.unableToOpenPath, SyllableDetectorConfig.swift:51 */
    SYLDET_ERR_PARSE_MISSING      = -21, /* ParseError.missingValue     :52                              */
    SYLDET_ERR_PARSE_INVALID      = -22, /* ParseError.invalidValue     :53                              */
    SYLDET_ERR_PARSE_LENGTH       = -23, /* ParseError.mismatchedLength :54                              */
    SYLDET_ERR_UNSUPPORTED        = -30
} syldet_status_t;

/* WindowType, CircularShortTimeFourierTransform.swift:12-29.  SyllableDetector always
 * selects hamming (SyllableDetector.swift:43); the others are the STFT class's options. */
typedef enum { SYLDET_WINDOW_NONE = 0, SYLDET_WINDOW_HAMMING = 1, SYLDET_WINDOW_HANNING = 2,
               SYLDET_WINDOW_BLACKMAN = 3 } syldet_window_t;
/* SyllableDetectorConfig.Scaling, SyllableDetectorConfig.swift:13-30 */
typedef enum { SYLDET_SCALING_LINEAR = 0, SYLDET_SCALING_LOG = 1, SYLDET_SCALING_DB = 2 } syldet_scaling_t;
/* extractPower (|X|, what the detector uses) vs extractMagnitude (|X|^2),
 * CircularShortTimeFourierTransform.swift:280 / :221 (the names are as in the reference) */
typedef enum { SYLDET_SPECTRUM_POWER = 0, SYLDET_SPECTRUM_MAGNITUDE = 1 } syldet_spectrum_t;
/* processing functions accepted by SyllableDetectorConfig.swift:133-151 (inputs) and
 * :158-167 (outputs: mapminmax, mapstd only)                                            */
typedef enum { SYLDET_FN_L2NORMALIZE = 0, SYLDET_FN_NORMALIZE = 1, SYLDET_FN_NORMALIZESTD = 2,
               SYLDET_FN_MAPMINMAX = 3, SYLDET_FN_MAPSTD = 4 } syldet_fn_kind_t;
/* transfer functions, SyllableDetectorConfig.swift:250-256 / NeuralNet.swift:185-228 */
typedef enum { SYLDET_TF_TANSIG = 0, SYLDET_TF_LOGSIG = 1, SYLDET_TF_PURELIN = 2,
               SYLDET_TF_SATLIN = 3 } syldet_transfer_t;
/* which outputs raise the detection flag: output 0 (SyllableDetector.lastDetected,
 * SyllableDetector.swift:27-31) or any output (CLI, TrackDetector.swift:72-77)          */
typedef enum { SYLDET_RULE_FIRST = 0, SYLDET_RULE_ANY = 1 } syldet_rule_t;

/* MapMinMax / MapStd parameters, NeuralNet.swift:111-182 (count = vector length; unused
 * and 0 for the parameter-free functions)                                               */
typedef struct {
    int32_t kind;            /* syldet_fn_kind_t */
    int32_t count;
    const float *x_offsets;
    const float *gains;
    float y;                 /* yMin (mapminmax) / yMean (mapstd) */
} syldet_fn_t;

/* NeuralNetLayer, NeuralNet.swift:329-378.  weights row-major [outputs][inputs]
 * (vDSP_mmul M=outputs, P=inputs at :368; convert_to_text.m:202).                        */
typedef struct {
    int32_t inputs, outputs;
    int32_t transfer;        /* syldet_transfer_t */
    const float *weights;
    const float *biases;
} syldet_layer_t;

/* SyllableDetectorConfig, SyllableDetectorConfig.swift:11-45 (+ the STFT options the
 * detector fixes).  All arrays are copied by syldet_create.                              */
typedef struct {
    double sampling_rate;            /* samplingRate  */
    int32_t fourier_length;          /* fourierLength */
    int32_t window_length;           /* windowLength  */
    int32_t window_overlap;          /* windowOverlap; negative = gap between windows */
    double freq_lo, freq_hi;         /* freqRange     */
    int32_t time_range;              /* timeRange     */
    int32_t scaling;                 /* syldet_scaling_t  */
    int32_t window;                  /* syldet_window_t; the detector uses SYLDET_WINDOW_HAMMING */
    int32_t spectrum;                /* syldet_spectrum_t; the detector uses SYLDET_SPECTRUM_POWER */
    int32_t rule;                    /* syldet_rule_t     */
    int32_t n_input_fns;
    const syldet_fn_t *input_fns;    /* net.inputProcessing, applied in order */
    int32_t n_layers;
    const syldet_layer_t *layers;    /* net.layers */
    int32_t n_output_fns;
    const syldet_fn_t *output_fns;   /* net.outputProcessing, applied in order as reverse maps */
    int32_t n_thresholds;
    const double *thresholds;        /* thresholds (Double) */
} syldet_config_t;

/* Derived geometry (what SyllableDetector.init computes, SyllableDetector.swift:42-60). */
typedef struct {
    int32_t gap, overlap, hop;       /* CircularShortTimeFourierTransform.swift:66-73; hop = gap + W - overlap */
    int32_t f0, f1;                  /* frequencyIndexRange, :166-191 */
    int32_t bins;                    /* F = f1 - f0 */
    int32_t inputs;                  /* F * timeRange == net.inputs */
    int32_t outputs;                 /* net.outputs */
    int32_t first_index;             /* sample number of evaluation 0, TrackDetector.swift:39-42 */
    int32_t engine;                  /* which kernel family create selected (syldet_engine_t) */
} syldet_geometry_t;

/* SYLDET_ENGINE_WIDE_BF16: two-layer networks with a wide hidden layer (BASELINE: 4096 units) evaluated as a bf16
 * MFMA GEMM over thousands of evaluations, fp32 accumulate.  Inputs and first-layer weights are rounded to bf16, so
 * results agree with the fp32 engines to ~1e-3, not 1e-5: opt-in only, never selected by AUTO.                   */
typedef enum { SYLDET_ENGINE_AUTO = 0, SYLDET_ENGINE_GENERIC = 1, SYLDET_ENGINE_FUSED = 2,
               SYLDET_ENGINE_WIDE_BF16 = 3 } syldet_engine_t;

typedef struct syldet syldet_t;

/* ---- library ---- */
int         syldet_abi_version(void);
const char *syldet_strerror(int status);
/* message of the last failing call on this thread (HIP error text, parse key, ...) */
const char *syldet_last_error(void);

/* ---- configuration file ----
 * SyllableDetectorConfig.init(fromTextFile:), SyllableDetectorConfig.swift:170-277.
 * On success *out owns every array it points to; free with syldet_config_free.
 * window/spectrum/rule are set to the detector's fixed choices (hamming, power, first). */
int  syldet_config_load_text(const char *path, syldet_config_t **out);
void syldet_config_free(syldet_config_t *cfg);
/* the same validation SyllableDetector.init performs, without touching a device */
int  syldet_config_geometry(const syldet_config_t *cfg, syldet_geometry_t *out);
/* CircularShortTimeFourierTransform.frequencyIndexRange, :166-191 (1 = range found, 0 = nil) */
int  syldet_frequency_index_range(int32_t fourier_length, double sampling_rate, double lo, double hi,
                                  int32_t *f0, int32_t *f1);
/* WindowType.createWindow, :19-28 */
int  syldet_make_window(int32_t window, int32_t length, float *out);

/* ---- detector bank ----
 * SyllableDetector.init(config:), SyllableDetector.swift:37-74, for n_channels
 * independent channels on HIP device `device`.  engine: SYLDET_ENGINE_AUTO unless a
 * test wants a specific kernel family.                                                  */
int syldet_create(const syldet_config_t *cfg, int32_t n_channels, int32_t device, int32_t engine,
                  syldet_t **out);
int syldet_destroy(syldet_t *h);
int syldet_get_geometry(const syldet_t *h, syldet_geometry_t *out);
int32_t syldet_channels(const syldet_t *h);

/* frames J = floor((S - gap - W)/hop) + 1 and evaluations E = J - T + 1 that S samples
 * per channel yield (extractPower's availability rule :286-288 + consume :299-302;
 * processNewValue's :164-178)                                                            */
int64_t syldet_count_frames(const syldet_t *h, int64_t n_samples);
int64_t syldet_count_evals(const syldet_t *h, int64_t n_samples);

/* ---- batch: the whole of `while detector.processNewValue() {...}` for every channel ----
 * (TrackDetector.swift:62-77 / Processor.swift:136-141).
 * samples  [C][channel_stride] fp32, the first n_samples of each row are used;
 * outputs  [C][E][outputs]    fp32  = lastOutputs after each evaluation;
 * flags    [C][E]             u8    = Double(out) >= threshold under cfg.rule;
 * Either output pointer may be NULL.                                                     */
int syldet_run_device(syldet_t *h, const float *d_samples, int64_t n_samples, int64_t channel_stride,
                      float *d_outputs, uint8_t *d_flags, void *hip_stream);
int syldet_run(syldet_t *h, const float *samples, int64_t n_samples, int64_t channel_stride,
               float *outputs, uint8_t *flags);

/* syldet_run cuts a long recording along time into stages of about 256 MiB of input and overlaps the H2D copy of the next
 * stage with the kernel of this one and the D2H copy of the last (device staging: two stages, whatever the length).
 * The copies read and write the caller's rows in place: buffers from syldet_host_alloc are page-locked, their copies
 * truly asynchronous; ordinary buffers work too (the runtime pins the pages of each copy as it goes).  TPCircularBufferInit (TPCircularBuffer.c:43-124) is the reference's
 * allocation of the buffer audio is produced into; this is its counterpart for a host that feeds a GPU.                 */
int syldet_host_alloc(size_t bytes, void **out);
int syldet_host_free(void *p);

/* the spectrogram columns the detector feeds its network, [C][J][bins] fp32
 * (processFourierData, SyllableDetector.swift:134-151; linear values, before scaling)    */
int syldet_spectrogram_device(syldet_t *h, const float *d_samples, int64_t n_samples, int64_t channel_stride,
                              float *d_columns, void *hip_stream);
int syldet_spectrogram(syldet_t *h, const float *samples, int64_t n_samples, int64_t channel_stride,
                       float *columns);

/* detection sample numbers with debounce, TrackDetector.swift:39-43,:65-100:
 * idx_e = first_index + e*hop; emit iff flag && debounce_until < idx_e, then
 * debounce_until = idx_e + Int(debounce_seconds * samplingRate).
 * indices [C][capacity] int64 (first counts[c] valid), counts [C] int64 (may exceed
 * capacity: the number that would have been written).                                    */
int syldet_detections_device(syldet_t *h, const uint8_t *d_flags, int64_t n_evals, double debounce_seconds,
                             int64_t *d_indices, int64_t capacity, int64_t *d_counts, void *hip_stream);
int syldet_detections(syldet_t *h, const uint8_t *flags, int64_t n_evals, double debounce_seconds,
                      int64_t *indices, int64_t capacity, int64_t *counts);

/* ---- measurement (replaces the reference's Time stopwatch, SyllableDetector/Time.swift:36-100,
 * which wraps processNewValue in ViewControllerSimulator.swift:309-319) ----
 * With profiling enabled every kernel of a batch call is bracketed by HIP events on the stream it
 * is launched on.  syldet_last_timings blocks until the last call's events have completed and
 * returns up to `capacity` kernel durations in milliseconds, in launch order, with their names.
 * The exact recomputation behind the fused kernels' precision guard ("fixup_kernel") is listed for
 * the calls that gave it work (syldet_fixup_stats' items > 0): on ordinary audio it is an empty launch.
 * Its duration is the kernel's own (device clock, first workgroup in to last out), and reading it
 * waits for the stream the call was made on.                                                       */
int syldet_profile(syldet_t *h, int enable);
int syldet_last_timings(syldet_t *h, double *milliseconds, const char **names, int32_t capacity, int32_t *count);
/* Keep the events of the last `calls` batch calls instead of one (waits for the handle's stream, drops what was recorded), so
 * that a measurement loop need not wait for every call before making the next; syldet_timings reads the call made
 * `calls_back` calls before the last one (0: the last), *count = 0 if that call is not held.                              */
int syldet_profile_history(syldet_t *h, int32_t calls);
int syldet_timings(syldet_t *h, int32_t calls_back, double *milliseconds, const char **names, int32_t capacity, int32_t *count);
/* The fused kernels compute on a block-floating-point grid.  The symmetric-fold kernel (the reference's example class: at
 * most 4 hidden units, no normaliser or l2normalize, windows of 64 / 128 / 192 / 256 samples) gives every FRAME its own
 * power-of-two scale: its results are a function of the samples under the window alone -- the same bits whether the audio
 * arrives through the streaming calls or a batch call, however it is tiled -- and only what no grid can hold (an infinite
 * sample, levels 2^45 apart inside one window) is recomputed.  The two older kernels (wider networks, normalize /
 * normalizestd chains) scale per 64 / 128-frame pass: evaluations whose windows that grid cannot hold to the 1e-5 contract --
 * a quiet stretch right behind a click, a level step of hundreds of dB -- are detected on the device and recomputed from
 * the samples in fp64 (and NaN, which the reference yields exactly for the windows that contain the offending sample:
 * NeuralNet.swift:47-59, appears exactly there); between two tilings of the same audio their results agree to a few 1e-7.  *items = 16-evaluation work items the last completed batch call of this handle recomputed
 * (0 for ordinary audio), *overflow = 1 if a work list was ever too small (never, by construction).  Blocks; call it after
 * the stream the batch call ran on has been synchronised.  No reference counterpart (diagnostic).                       */
int syldet_fixup_stats(syldet_t *h, int64_t *items, int32_t *overflow);
/* How the batch call tiles a channel for n_samples per channel: evaluations per workgroup segment of the fused kernels
 * (consecutive segments of a channel are computed by different workgroups; 0: the engine in use has no such seams).
 * Results do not depend on it; verification uses it to aim spot checks at the seams.  No reference counterpart.          */
int64_t syldet_segment_evals(const syldet_t *h, int64_t n_samples);

/* ---- streaming: the reference's per-detector API, one call per channel ----
 * Each channel owns a single-producer / single-consumer sample ring like the reference's
 * TPCircularBuffer (TPCircularBuffer.h:14,102-189): append* never locks or allocates and may run
 * on an audio I/O thread (AudioInterface.swift:67-70 -> Processor.swift:124) while another thread
 * processes.  appendAudioData(_:withSamples:), SyllableDetector.swift:129-132              */
int syldet_append(syldet_t *h, int32_t channel, const float *data, int64_t n_samples);
/* appendInterleavedData(_:withSamples:fromChannel:ofTotalChannels:),
 * CircularShortTimeFourierTransform.swift:203-217: de-interleaves frame-major audio into
 * every channel of the bank (total_channels == syldet_channels(h))                       */
int syldet_append_interleaved(syldet_t *h, const float *data, int64_t n_frames, int32_t total_channels);
/* The same call's fromChannel: the bank sits on a SUBSET of a wider device stream -- channel c of the bank takes channel
 * source_channel[c] of the total_channels interleaved in `data` (appendInterleavedData(_:withSamples:fromChannel:ofTotalChannels:),
 * CircularShortTimeFourierTransform.swift:203-217, takes one channel of the stream per call: :213's stride is the stream's width).
 * source_channel: syldet_channels(h) entries, each in [0, total_channels); a stream channel may feed several bank channels.    */
int syldet_append_interleaved_channels(syldet_t *h, const float *data, int64_t n_frames, int32_t total_channels,
                                       const int32_t *source_channel);
/* processNewValue() -> Bool, SyllableDetector.swift:153-217: 1 = a new evaluation is in
 * last_outputs, 0 = not enough data yet                                                  */
int syldet_process_new_value(syldet_t *h, int32_t channel);
/* The consumer loop of a multi-channel Processor (`for d in detectors { while d.processNewValue() … }`,
 * Processor.swift:128-141) in one device round trip: evaluates everything every channel has pending
 * (one staged copy + one launch per distinct evaluation count, i.e. one when the channels are fed
 * together) and queues the results; the following syldet_process_new_value calls hand them out one by
 * one without touching the device.  *n_queued (optional) = evaluations added over all channels.      */
int syldet_process_all(syldet_t *h, int64_t *n_queued);
/* evaluations computed and not yet handed out by syldet_process_new_value                */
int64_t syldet_pending_evaluations(const syldet_t *h, int32_t channel);
/* lastOutputs, :26 (zeros before the first evaluation, :70)                              */
int syldet_last_outputs(const syldet_t *h, int32_t channel, float *out);
/* lastDetected, :27-31                                                                   */
int syldet_last_detected(const syldet_t *h, int32_t channel);
/* seenSyllable(), :220-230: drains every pending evaluation, 1 if any was detected       */
int syldet_seen_syllable(syldet_t *h, int32_t channel);

/* ---- ingest: the steps immediately before the path ----
 * Frame-major (interleaved) audio, as a decoder or a multi-channel device delivers it:
 * appendInterleavedData(_:withSamples:fromChannel:ofTotalChannels:),
 * CircularShortTimeFourierTransform.swift:203-217, for every channel at once.
 * interleaved [n_frames][total_channels] fp32 -> rows of channels first_channel ..
 * first_channel + n_channels - 1 in out [n_channels][out_stride].                        */
int syldet_deinterleave_device(const float *d_interleaved, int64_t n_frames, int32_t total_channels,
                               int32_t first_channel, int32_t n_channels, float *d_out, int64_t out_stride,
                               void *hip_stream);
/* The batch call on interleaved audio (total_channels == syldet_channels(h)): de-interleave
 * on the device, then exactly syldet_run_device / syldet_run.                             */
int syldet_run_interleaved_device(syldet_t *h, const float *d_interleaved, int64_t n_frames, int32_t total_channels,
                                  float *d_outputs, uint8_t *d_flags, void *hip_stream);
int syldet_run_interleaved(syldet_t *h, const float *interleaved, int64_t n_frames, int32_t total_channels,
                           float *outputs, uint8_t *flags);

/* ---- the exchange step of the multi-GPU path (no reference counterpart: the reference runs one process) ----
 * Channels shard across GPUs with no data-path collective; the one exchange is the gather of the detection
 * flags, and it travels as bits: bit b of byte t of a row = flag 8 t + b, rows padded to whole bytes
 * ((row_len + 7) / 8 bytes per row).  Device pointers (d_flags of the unpack 8-byte aligned), asynchronous on
 * `hip_stream`; rows <= 65535.                                                                            */
int syldet_pack_flags_device(const uint8_t *d_flags, int64_t rows, int64_t row_len, uint8_t *d_bits, void *hip_stream);
int syldet_unpack_flags_device(const uint8_t *d_bits, int64_t rows, int64_t row_len, uint8_t *d_flags, void *hip_stream);

/* ---- one bank over several GPUs, ONE process ----
 * The reference is one process that owns every channel: Processor.swift:57-59 builds one SyllableDetector per channel and
 * one serial queue drains them all (:82, :128-141); main.swift:86-89 builds one TrackDetector per track and one loop runs
 * them (:126-130).  A sharded bank keeps that shape for a host with several MI355X: one handle, one call per batch; the
 * library places a sub-bank and a stream on every listed device, splits the channels into contiguous blocks (the first
 * n_channels % n_devices devices take one more), and -- with fewer channels than devices -- splits a channel's TIME axis
 * instead: a shard then computes a contiguous range of one channel's evaluations from its samples plus a halo of
 * (timeRange - 1) hop + window - hop (+ gap) samples (evaluation e is frames e .. e + timeRange - 1, frame j is samples
 * [j hop + gap, j hop + gap + window): SyllableDetector.swift:153-217, CircularShortTimeFourierTransform.swift:286-302).
 * The kernels AUTO selects for the benchmark configurations scale per frame (fold kernel) or per hop-aligned block
 * (block-transform, FFT kernels), so a shard's results there are the unsharded bank's bit for bit; the pass-scaled fused
 * kernels (SYLDET_FUSED_NOFOLD / _CLASSIC, shapes outside the fold kernel's class) agree between tilings to a few 1e-7 only.
 * The data path has no collective.  The one exchange -- every device receives every channel's detection flags -- is ONE
 * all-gather of the bit-packed rows per batch, on RCCL communicators the library makes itself (ncclCommInitAll, one
 * process; librccl is loaded on first use, so hosts with one GPU never pay for it).                                   */
typedef struct syldet_sharded syldet_sharded_t;

typedef struct {
    int32_t device;              /* HIP device of the shard                                                */
    int32_t first_channel;       /* the shard owns channels [first_channel, first_channel + channels)      */
    int32_t channels;
    int32_t part, parts;         /* its place along the time axis of its channel (0 of 1 unless n_channels < n_devices) */
} syldet_shard_t;

/* how the flags travel between the devices of a sharded bank */
typedef enum {
    SYLDET_EXCHANGE_RCCL = 0,        /* one ncclAllGather per device inside one ncclGroupStart/End (xGMI)            */
    SYLDET_EXCHANGE_PEER_COPY = 1    /* hipMemcpyPeerAsync of every shard's packed rows to every device: no RCCL in
                                        the process; also what a bank with one device listed twice uses (a rehearsal
                                        of the shard logic on a one-GPU box: RCCL refuses duplicate devices)        */
} syldet_exchange_t;

/* The shard table alone (host arithmetic, no device): out[i] for i < n_shards, device = i. */
int syldet_shard_table(int32_t n_channels, int32_t n_shards, syldet_shard_t *out);
/* Evaluations [*first, *first + *count) of a channel with n_evals evaluations that part `part` of `parts` computes, and the
 * samples [*s0, *s1) of the recording it reads for them (cfg gives hop, gap, window, timeRange).                         */
int syldet_shard_evaluations(int64_t n_evals, int32_t parts, int32_t part, int64_t *first, int64_t *count);
int syldet_shard_samples(const syldet_config_t *cfg, int64_t first_eval, int64_t count, int64_t *s0, int64_t *s1);

int syldet_create_sharded(const syldet_config_t *cfg, int32_t n_channels, const int32_t *devices, int32_t n_devices,
                          int32_t engine, int32_t exchange, syldet_sharded_t **out);
int syldet_sharded_destroy(syldet_sharded_t *b);
int32_t syldet_sharded_channels(const syldet_sharded_t *b);
int32_t syldet_sharded_shards(const syldet_sharded_t *b);
int syldet_sharded_shard(const syldet_sharded_t *b, int32_t shard, syldet_shard_t *out);
/* the shard's own bank (borrowed: destroyed with the sharded bank), the stream its kernels (and own results) are queued on,
 * and the stream its share of the exchange runs on (d_flags_all[shard] is complete when THAT stream has drained: the exchange
 * of batch i runs beside the kernels of batch i + 1, as Processor.swift:128-141's queue hands out results while audio arrives) */
syldet_t *syldet_sharded_bank(syldet_sharded_t *b, int32_t shard);
void *syldet_sharded_stream(syldet_sharded_t *b, int32_t shard);
void *syldet_sharded_exchange_stream(syldet_sharded_t *b, int32_t shard);
/* For a recording of n_samples per channel: the samples [*s0, *s1) shard `shard` reads of each of its channels (all of them
 * unless time-sharded) and the evaluations [*e0, *e0 + *count) it computes.  Any output pointer may be NULL.            */
int syldet_sharded_ranges(const syldet_sharded_t *b, int32_t shard, int64_t n_samples, int64_t *s0, int64_t *s1,
                          int64_t *e0, int64_t *count);
/* Host buffers, the whole bank in one call: samples [C][channel_stride] -> outputs [C][E][outputs], flags [C][E], as
 * syldet_run.  Every shard's copies and kernels are queued before any is waited for (one pipelined H2D / kernel / D2H
 * chain per device); results land in the caller's rows directly, so this form needs no collective.  Blocks.              */
int syldet_sharded_run(syldet_sharded_t *b, const float *samples, int64_t n_samples, int64_t channel_stride,
                       float *outputs, uint8_t *flags);
/* Device buffers: d_samples[i] is shard i's block on its device -- [channels_i][strides[i]] rows holding the shard's
 * sample range (syldet_sharded_ranges) of a recording of n_samples per channel; d_outputs[i] [channels_i][count_i][outputs]
 * and d_flags[i] [channels_i][count_i] receive its own results (either array, or any entry, may be NULL);
 * d_flags_all[i], when the array is given, receives EVERY channel's flags [C][E] on device i (8-byte aligned) through the
 * one exchange.  The bank's streams are its own (hipStreamNonBlocking): work the caller has queued elsewhere that produces the
 * samples -- or still reads memory now handed over as a result array -- must have finished (or be ordered by the caller's own
 * events on syldet_sharded_stream) before the call.  Asynchronous: every shard's kernel is launched before the exchange is queued, and the exchange runs on
 * streams of its own (two sets of buffers in turn), so the next call's kernels start without waiting for this call's
 * collective.  The queueing itself runs on the bank's launcher threads, every shard at once (syldet_sharded_launcher_threads);
 * the call returns when every shard's work is queued.  Results are complete after syldet_sharded_synchronize (or after synchronising syldet_sharded_stream(b, i) for
 * shard i's own results, syldet_sharded_exchange_stream(b, i) for d_flags_all[i]).                                       */
int syldet_sharded_run_device(syldet_sharded_t *b, const float *const *d_samples, int64_t n_samples, const int64_t *strides,
                              float *const *d_outputs, uint8_t *const *d_flags, uint8_t *const *d_flags_all);
int syldet_sharded_synchronize(syldet_sharded_t *b);
/* RCCL ranks behind the exchange (0 under SYLDET_EXCHANGE_PEER_COPY) */
int32_t syldet_sharded_rccl_ranks(const syldet_sharded_t *b);
/* Brings the exchange up NOW instead of inside the first gathering batch: under SYLDET_EXCHANGE_RCCL loads librccl and makes
 * the communicators (ncclCommInitAll over the bank's devices); under the copy exchange enables peer access between the bank's
 * devices where the hardware offers it.  A caller that wants to fall back (a host whose RCCL does not come up) calls this
 * right after syldet_create_sharded and, on an error, destroys the bank and makes it again with SYLDET_EXCHANGE_PEER_COPY
 * -- in the same process: nothing here needs a fresh one.                                                                   */
int syldet_sharded_connect(syldet_sharded_t *b);
/* Launcher threads of the bank: one per shard, alive as long as the bank, each with its shard's device current; a batch call's
 * per-shard queueing (kernel, packing, the exchange's waits and records, unpacking) runs on all of them at once.  0 for a bank
 * of one shard and for banks made under SYLDET_SHARDED_INLINE=1, whose calls queue shard after shard on the caller's thread
 * (the reference's own shape: one serial queue, Processor.swift:82).                                                         */
int32_t syldet_sharded_launcher_threads(const syldet_sharded_t *b);

/* ResamplerLinear, Common/Resampler.swift:20-76 (used when the device rate differs from the
 * network's: Processor.swift:116-121, ViewControllerProcessor.swift:247-250), for n_channels
 * independent streams fed in lock-step.  Stateful like the reference: the fractional
 * position (`offset`) and the last input sample of every channel carry over to the next
 * call.  Results are bit-identical to the reference's arithmetic order (fp32, no FMA).    */
typedef struct syldet_resampler syldet_resampler_t;
int syldet_resampler_create(double rate_in, double rate_out, int32_t n_channels, int32_t device,
                            syldet_resampler_t **out);
int syldet_resampler_destroy(syldet_resampler_t *r);
/* samples per channel the next call produces from n_in input samples (:40)               */
int64_t syldet_resampler_count(const syldet_resampler_t *r, int64_t n_in);
/* resampleVector(_:ofLength:), :36-69.  in [C][in_stride] -> out [C][out_stride]; *n_out
 * (host) receives the per-channel output length (= syldet_resampler_count before the call) */
int syldet_resample_device(syldet_resampler_t *r, const float *d_in, int64_t n_in, int64_t in_stride, float *d_out,
                           int64_t out_stride, int64_t *n_out, void *hip_stream);
int syldet_resample(syldet_resampler_t *r, const float *in, int64_t n_in, int64_t in_stride, float *out,
                    int64_t out_stride, int64_t *n_out);

/* Whole-recording rate conversion for offline input.  The reference's command line tool never resamples itself: it asks
 * AVFoundation to deliver every track at the network's rate (audioSettings, SyllableDetector.swift:19-23, handed to
 * AVAssetReaderTrackOutput at TrackDetector.swift:35).  This is that step for a decoded file: output sample i is the linear
 * interpolation of the input at position i * rate_in / rate_out, the position computed in fp64 (ResamplerLinear above is a
 * streaming object for short live buffers: its fp32 position ramp and its buffer carry are not meant for minutes of audio
 * in one call).  Stateless.  in [C][in_stride] -> out [C][out_stride]; *n_out = syldet_convert_rate_count(n_in, ...).    */
int64_t syldet_convert_rate_count(int64_t n_in, double rate_in, double rate_out);
int syldet_convert_rate_device(const float *d_in, int64_t n_in, int64_t in_stride, int32_t n_channels, double rate_in,
                               double rate_out, float *d_out, int64_t out_stride, int64_t *n_out, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* SYLDET_H */
