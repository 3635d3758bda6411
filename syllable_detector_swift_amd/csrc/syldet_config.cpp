// syldet_config.cpp -- host-only part of libsyldet: status text, configuration ownership,
// the validation SyllableDetector.init performs, and the `key = value` text format of
// SyllableDetectorConfig.init(fromTextFile:).  No device code here.
//
// Reference citations are relative to the reference root.

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

#include "syldet_internal.hpp"

namespace sd {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }
int fail(int status, const std::string &msg)
{
    g_last_error = msg;
    return status;
}

// ---------------------------------------------------------------- ownership

static int copy_fn(const syldet_fn_t &src, syldet_fn_t &dst, std::vector<float> &xo, std::vector<float> &ga)
{
    dst = src;
    const bool has_params = src.kind == SYLDET_FN_MAPMINMAX || src.kind == SYLDET_FN_MAPSTD;
    if (src.kind < SYLDET_FN_L2NORMALIZE || src.kind > SYLDET_FN_MAPSTD)
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "unknown processing function kind");
    if (has_params) {
        if (src.count <= 0 || !src.x_offsets || !src.gains)
            return fail(SYLDET_ERR_INVALID_ARGUMENT, "mapminmax/mapstd need x_offsets and gains");
        xo.assign(src.x_offsets, src.x_offsets + src.count);
        ga.assign(src.gains, src.gains + src.count);
    } else {
        xo.clear();
        ga.clear();
        dst.count = 0;
    }
    return SYLDET_OK;
}

int OwnedConfig::assign(const syldet_config_t &src)
{
    view = src;
    if (src.n_input_fns < 0 || src.n_output_fns < 0 || src.n_layers < 0 || src.n_thresholds < 0)
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "negative count in configuration");
    if ((src.n_input_fns && !src.input_fns) || (src.n_output_fns && !src.output_fns) ||
        (src.n_layers && !src.layers) || (src.n_thresholds && !src.thresholds))
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL array in configuration");
    input_fns.resize(src.n_input_fns);
    fn_xoff_in.resize(src.n_input_fns);
    fn_gain_in.resize(src.n_input_fns);
    for (int i = 0; i < src.n_input_fns; i++)
        if (int st = copy_fn(src.input_fns[i], input_fns[i], fn_xoff_in[i], fn_gain_in[i])) return st;
    output_fns.resize(src.n_output_fns);
    fn_xoff_out.resize(src.n_output_fns);
    fn_gain_out.resize(src.n_output_fns);
    for (int i = 0; i < src.n_output_fns; i++) {
        // outputs accept mapminmax / mapstd only (SyllableDetectorConfig.swift:158-167)
        if (src.output_fns[i].kind != SYLDET_FN_MAPMINMAX && src.output_fns[i].kind != SYLDET_FN_MAPSTD)
            return fail(SYLDET_ERR_INVALID_ARGUMENT, "output processing accepts mapminmax and mapstd only");
        if (int st = copy_fn(src.output_fns[i], output_fns[i], fn_xoff_out[i], fn_gain_out[i])) return st;
    }
    layers.resize(src.n_layers);
    weights.resize(src.n_layers);
    biases.resize(src.n_layers);
    for (int l = 0; l < src.n_layers; l++) {
        const syldet_layer_t &L = src.layers[l];
        // NeuralNetLayer.init guards, NeuralNet.swift:340-349
        if (L.inputs <= 0 || L.outputs <= 0)
            return fail(SYLDET_ERR_LAYER_SHAPE, "Each layer must have at least one input and at least one output.");
        if (!L.weights || !L.biases) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL layer weights/biases");
        if (L.transfer < SYLDET_TF_TANSIG || L.transfer > SYLDET_TF_SATLIN)
            return fail(SYLDET_ERR_INVALID_ARGUMENT, "unknown transfer function");
        layers[l] = L;
        weights[l].assign(L.weights, L.weights + (size_t)L.inputs * (size_t)L.outputs);
        biases[l].assign(L.biases, L.biases + L.outputs);
    }
    thresholds.assign(src.thresholds, src.thresholds + src.n_thresholds);
    relink();
    return SYLDET_OK;
}

void OwnedConfig::relink()
{
    for (size_t i = 0; i < input_fns.size(); i++) {
        input_fns[i].x_offsets = fn_xoff_in[i].empty() ? nullptr : fn_xoff_in[i].data();
        input_fns[i].gains = fn_gain_in[i].empty() ? nullptr : fn_gain_in[i].data();
    }
    for (size_t i = 0; i < output_fns.size(); i++) {
        output_fns[i].x_offsets = fn_xoff_out[i].empty() ? nullptr : fn_xoff_out[i].data();
        output_fns[i].gains = fn_gain_out[i].empty() ? nullptr : fn_gain_out[i].data();
    }
    for (size_t l = 0; l < layers.size(); l++) {
        layers[l].weights = weights[l].data();
        layers[l].biases = biases[l].data();
    }
    view.n_input_fns = (int32_t)input_fns.size();
    view.input_fns = input_fns.empty() ? nullptr : input_fns.data();
    view.n_output_fns = (int32_t)output_fns.size();
    view.output_fns = output_fns.empty() ? nullptr : output_fns.data();
    view.n_layers = (int32_t)layers.size();
    view.layers = layers.empty() ? nullptr : layers.data();
    view.n_thresholds = (int32_t)thresholds.size();
    view.thresholds = thresholds.empty() ? nullptr : thresholds.data();
}

// ---------------------------------------------------------------- geometry

static bool is_pow2(int64_t v) { return v != 0 && (v & (v - 1)) == 0; }   // Common.swift:26-30

static int frequency_index_range(int N, double fs, double lo, double hi, int32_t *f0, int32_t *f1)
{
    // CircularShortTimeFourierTransform.frequencyIndexRange, :166-191
    if (!(lo >= 0.0 && hi > lo)) return 0;
    const int half = N / 2;
    const double from_frequency = (double)N / fs;
    const int start = (int)std::ceil(from_frequency * lo);
    if (start >= half) return 0;
    int end = (int)std::floor(from_frequency * hi) + 1;
    if (end < start) return 0;
    if (end > half) end = half;
    *f0 = start;
    *f1 = end;
    return 1;
}

int compute_geometry(const syldet_config_t &c, syldet_geometry_t *g)
{
    std::memset(g, 0, sizeof(*g));
    const int N = c.fourier_length, W = c.window_length, ov = c.window_overlap;
    // CircularShortTimeFourierTransform.init :76-88
    if (W <= 0) return fail(SYLDET_ERR_INVALID_ARGUMENT, "windowLength must be positive");
    if (ov >= W) return fail(SYLDET_ERR_OVERLAP, "Invalid overlap value.");
    if (N <= 0 || !is_pow2(N)) return fail(SYLDET_ERR_FFT_SIZE, "The FFT size must be a power of 2.");
    if (W > N) return fail(SYLDET_ERR_FFT_SIZE, "The FFT size must be greater than or equal to the window length.");
    if (N < 4) return fail(SYLDET_ERR_FFT_SIZE, "The FFT size must be at least 4.");
    if (!(c.sampling_rate > 0.0)) return fail(SYLDET_ERR_INVALID_ARGUMENT, "samplingRate must be positive");
    g->gap = ov < 0 ? -ov : 0;                     // :66-73
    g->overlap = ov < 0 ? 0 : ov;
    g->hop = g->gap + W - g->overlap;              // consumed per frame :299-302
    if (!frequency_index_range(N, c.sampling_rate, c.freq_lo, c.freq_hi, &g->f0, &g->f1))
        return fail(SYLDET_ERR_FREQ_RANGE, "The frequency range is invalid.");   // SyllableDetector.swift:46-48
    g->bins = g->f1 - g->f0;
    if (c.time_range < 1) return fail(SYLDET_ERR_INVALID_ARGUMENT, "timeRange must be at least 1");
    // NeuralNet.init :244-254
    if (c.n_layers < 1) return fail(SYLDET_ERR_LAYER_SHAPE, "Neural network must have 1 or more layers.");
    for (int l = 0; l < c.n_layers; l++) {
        if (c.layers[l].inputs <= 0 || c.layers[l].outputs <= 0)
            return fail(SYLDET_ERR_LAYER_SHAPE, "Each layer must have at least one input and at least one output.");
        if (l > 0 && c.layers[l - 1].outputs != c.layers[l].inputs)
            return fail(SYLDET_ERR_LAYER_SHAPE, "Number of inputs for layer " + std::to_string(l) + " does not match previous outputs.");
    }
    g->inputs = g->bins * c.time_range;
    g->outputs = c.layers[c.n_layers - 1].outputs;
    if (g->inputs != c.layers[0].inputs)           // SyllableDetector.swift:52-55
        return fail(SYLDET_ERR_INPUT_MISMATCH, "The neural network has " + std::to_string(c.layers[0].inputs) +
                    " inputs, but the configuration settings suggest there should be " + std::to_string(g->inputs) + ".");
    if (c.n_thresholds != g->outputs)              // SyllableDetector.swift:58-60
        return fail(SYLDET_ERR_THRESHOLD_MISMATCH, "The neural network has " + std::to_string(g->outputs) +
                    " outputs, but the configuration settings suggest there should be " + std::to_string(c.n_thresholds) + ".");
    // processing-function vector lengths (parseMapMinMax withCount, SyllableDetectorConfig.swift:114-126)
    for (int i = 0; i < c.n_input_fns; i++) {
        const syldet_fn_t &f = c.input_fns[i];
        if ((f.kind == SYLDET_FN_MAPMINMAX || f.kind == SYLDET_FN_MAPSTD) && f.count != g->inputs)
            return fail(SYLDET_ERR_INPUT_MISMATCH, "input processing vector length does not match net.inputs");
    }
    for (int i = 0; i < c.n_output_fns; i++)
        if (c.output_fns[i].count != g->outputs)
            return fail(SYLDET_ERR_INPUT_MISMATCH, "output processing vector length does not match net.outputs");
    if (c.scaling < SYLDET_SCALING_LINEAR || c.scaling > SYLDET_SCALING_DB ||
        c.window < SYLDET_WINDOW_NONE || c.window > SYLDET_WINDOW_BLACKMAN ||
        c.spectrum < SYLDET_SPECTRUM_POWER || c.spectrum > SYLDET_SPECTRUM_MAGNITUDE ||
        c.rule < SYLDET_RULE_FIRST || c.rule > SYLDET_RULE_ANY)
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad enum value in configuration");
    // TrackDetector.init :39-42
    int64_t first = (int64_t)W + (int64_t)(W - ov) * (c.time_range - 1);
    if (ov < 0) first -= ov;
    g->first_index = (int32_t)first;
    g->engine = SYLDET_ENGINE_AUTO;
    return SYLDET_OK;
}

void make_window(int window, int length, float *out)
{
    // WindowType.createWindow, CircularShortTimeFourierTransform.swift:19-28.
    // vDSP_hamm_window / vDSP_hann_window (denormalised) / vDSP_blkman_window, flag 0:
    // full-length table with denominator `length`; evaluated in double, stored as float.
    const double two_pi = 6.283185307179586476925286766559;
    for (int n = 0; n < length; n++) {
        const double a = two_pi * (double)n / (double)length;
        double v = 1.0;
        switch (window) {
        case SYLDET_WINDOW_HAMMING: v = 0.54 - 0.46 * std::cos(a); break;
        case SYLDET_WINDOW_HANNING: v = 0.5 * (1.0 - std::cos(a)); break;
        case SYLDET_WINDOW_BLACKMAN: v = 0.42 - 0.5 * std::cos(a) + 0.08 * std::cos(2.0 * a); break;
        default: break;
        }
        out[n] = (float)v;
    }
}

// ---------------------------------------------------------------- text format

namespace {

// String.trim(), Common.swift:17-19 (CharacterSet.whitespacesAndNewlines)
std::string trim(const std::string &s)
{
    const char *ws = " \t\n\r\f\v";
    const size_t a = s.find_first_not_of(ws);
    if (a == std::string::npos) return std::string();
    const size_t b = s.find_last_not_of(ws);
    return s.substr(a, b - a + 1);
}

// String.splitAtCharacter, Common.swift:21-23: Swift's split drops empty subsequences.
std::vector<std::string> split_at(const std::string &s, char c)
{
    std::vector<std::string> parts;
    size_t i = 0;
    while (i <= s.size()) {
        size_t j = s.find(c, i);
        if (j == std::string::npos) j = s.size();
        if (j > i) parts.push_back(s.substr(i, j - i));
        i = j + 1;
    }
    return parts;
}

using Dict = std::map<std::string, std::string>;

struct ParseFailure {
    int status;
    std::string text;
};

[[noreturn]] void throw_missing(const std::string &k) { throw ParseFailure{SYLDET_ERR_PARSE_MISSING, "missingValue(\"" + k + "\")"}; }
[[noreturn]] void throw_invalid(const std::string &k) { throw ParseFailure{SYLDET_ERR_PARSE_INVALID, "invalidValue(\"" + k + "\")"}; }
[[noreturn]] void throw_length(const std::string &k) { throw ParseFailure{SYLDET_ERR_PARSE_LENGTH, "mismatchedLength(\"" + k + "\")"}; }

// Swift's Double(String)/Float(String): the whole string must be one number, no
// surrounding whitespace.
bool to_double(const std::string &v, double *out)
{
    if (v.empty() || std::isspace((unsigned char)v.front())) return false;
    char *end = nullptr;
    errno = 0;
    const double d = std::strtod(v.c_str(), &end);
    if (end == v.c_str() || *end != '\0') return false;
    *out = d;
    return true;
}
bool to_float(const std::string &v, float *out)
{
    if (v.empty() || std::isspace((unsigned char)v.front())) return false;
    char *end = nullptr;
    errno = 0;
    const float f = std::strtof(v.c_str(), &end);   // rounds the decimal straight to float, as Float(String) does
    if (end == v.c_str() || *end != '\0') return false;
    *out = f;
    return true;
}
// Swift's Int(String): optional sign, decimal digits only.
bool to_int(const std::string &v, long long *out)
{
    size_t i = 0;
    if (v.empty()) return false;
    if (v[0] == '+' || v[0] == '-') i = 1;
    if (i >= v.size()) return false;
    for (size_t k = i; k < v.size(); k++)
        if (v[k] < '0' || v[k] > '9') return false;
    errno = 0;
    const long long x = std::strtoll(v.c_str(), nullptr, 10);
    if (errno == ERANGE) return false;
    *out = x;
    return true;
}

const std::string &lookup(const Dict &d, const std::string &k)
{
    auto it = d.find(k);
    if (it == d.end()) throw_missing(k);
    return it->second;
}
double parse_double(const std::string &k, const Dict &d)    // parseDouble :62-66
{
    double v;
    if (!to_double(lookup(d, k), &v)) throw_invalid(k);
    return v;
}
float parse_float(const std::string &k, const Dict &d)      // parseFloat :68-72
{
    float v;
    if (!to_float(lookup(d, k), &v)) throw_invalid(k);
    return v;
}
long long parse_int(const std::string &k, const Dict &d)    // parseInt :74-78
{
    long long v;
    if (!to_int(lookup(d, k), &v)) throw_invalid(k);
    return v;
}
std::vector<double> parse_double_array(const std::string &k, long long cnt, const Dict &d)   // :80-96
{
    const std::vector<std::string> parts = split_at(lookup(d, k), ',');
    std::vector<double> out;
    for (const std::string &p : parts) {
        double v;
        if (!to_double(trim(p), &v)) throw_invalid(k);
        out.push_back(v);
    }
    if (cnt >= 0 && (long long)out.size() != cnt) throw_length(k);
    return out;
}
std::vector<float> parse_float_array(const std::string &k, long long cnt, const Dict &d)     // :98-112
{
    const std::vector<std::string> parts = split_at(lookup(d, k), ',');
    std::vector<float> out;
    for (const std::string &p : parts) {
        float v;
        if (!to_float(trim(p), &v)) throw_invalid(k);
        out.push_back(v);
    }
    if ((long long)out.size() != cnt) throw_length(k);
    return out;
}

struct ParsedFn {
    int kind;
    std::vector<float> xoff, gain;
    float y = 0.0f;
};

ParsedFn parse_map(const std::string &nm, int kind, long long cnt, const Dict &d)   // parseMapMinMax / parseMapStd :114-126
{
    ParsedFn f;
    f.kind = kind;
    f.xoff = parse_float_array(nm + ".xOffsets", cnt, d);
    f.gain = parse_float_array(nm + ".gains", cnt, d);
    f.y = parse_float(nm + (kind == SYLDET_FN_MAPMINMAX ? ".yMin" : ".yMean"), d);
    return f;
}

ParsedFn parse_input_fn(const std::string &nm, long long cnt, const Dict &d)        // :128-156
{
    const std::string &fn = lookup(d, nm + ".function");
    if (fn == "mapminmax") return parse_map(nm, SYLDET_FN_MAPMINMAX, cnt, d);
    if (fn == "mapstd") return parse_map(nm, SYLDET_FN_MAPSTD, cnt, d);
    ParsedFn f;
    if (fn == "l2normalize") f.kind = SYLDET_FN_L2NORMALIZE;
    else if (fn == "normalize") f.kind = SYLDET_FN_NORMALIZE;
    else if (fn == "normalizestd") f.kind = SYLDET_FN_NORMALIZESTD;
    else throw_invalid(nm + ".function");
    return f;
}

ParsedFn parse_output_fn(const std::string &nm, long long cnt, const Dict &d)       // :158-168
{
    const std::string &fn = lookup(d, nm + ".function");
    if (fn == "mapminmax") return parse_map(nm, SYLDET_FN_MAPMINMAX, cnt, d);
    if (fn == "mapstd") return parse_map(nm, SYLDET_FN_MAPSTD, cnt, d);
    throw_invalid(nm + ".function");
}

struct Loaded {
    syldet_config_t view;   // first member: the pointer handed to the caller
    OwnedConfig *owned;
};

}  // namespace

static int load_text(const char *path, OwnedConfig &oc)
{
    // StreamReader(path:) :172-174
    std::ifstream in(path, std::ios::binary);
    if (!in) throw ParseFailure{SYLDET_ERR_PARSE_OPEN, std::string("unableToOpenPath(\"") + path + "\")"};
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string text = ss.str();

    // line loop :183-189: keep lines that split at "=" into exactly two non-empty parts
    Dict data;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t nl = text.find('\n', pos);
        if (nl == std::string::npos) nl = text.size();
        const std::string line = text.substr(pos, nl - pos);
        pos = nl + 1;
        const std::vector<std::string> parts = split_at(line, '=');
        if (parts.size() == 2) data[trim(parts[0])] = trim(parts[1]);
    }

    syldet_config_t &v = oc.view;
    std::memset(&v, 0, sizeof(v));
    v.sampling_rate = parse_double("samplingRate", data);                       // :195
    const long long flen = parse_int("fourierLength", data);                    // :198-201
    if (!is_pow2(flen) || flen > (1 << 30)) throw_invalid("fourierLength");
    v.fourier_length = (int32_t)flen;
    if (data.find("windowLength") == data.end()) v.window_length = v.fourier_length;   // :204-209
    else v.window_length = (int32_t)parse_int("windowLength", data);
    v.window_overlap = (int32_t)parse_int("windowOverlap", data);               // :212
    const std::vector<double> fr = parse_double_array("freqRange", 2, data);    // :215-217
    v.freq_lo = fr[0];
    v.freq_hi = fr[1];
    v.time_range = (int32_t)parse_int("timeRange", data);                       // :220
    try {                                                                       // :223-229
        oc.thresholds = parse_double_array("thresholds", -1, data);
    } catch (const ParseFailure &) {
        oc.thresholds = parse_double_array("threshold", -1, data);              // backwards compatibility
    }
    const std::string &sc = lookup(data, "scaling");                            // :232-237
    if (sc == "linear") v.scaling = SYLDET_SCALING_LINEAR;
    else if (sc == "log") v.scaling = SYLDET_SCALING_LOG;
    else if (sc == "db") v.scaling = SYLDET_SCALING_DB;
    else throw_invalid("scaling");

    const long long layer_count = parse_int("layers", data);                    // :240
    if (layer_count < 0 || layer_count > 4096) throw_invalid("layers");
    oc.layers.resize((size_t)layer_count);
    oc.weights.resize((size_t)layer_count);
    oc.biases.resize((size_t)layer_count);
    for (long long i = 0; i < layer_count; i++) {                               // :241-259
        const std::string nm = "layer" + std::to_string(i);
        const long long inputs = parse_int(nm + ".inputs", data);
        const long long outputs = parse_int(nm + ".outputs", data);
        if (inputs < 0 || outputs < 0 || inputs > (1 << 24) || outputs > (1 << 24)) throw_invalid(nm + ".inputs");
        oc.weights[i] = parse_float_array(nm + ".weights", inputs * outputs, data);
        oc.biases[i] = parse_float_array(nm + ".biases", outputs, data);
        const std::string &tf = lookup(data, nm + ".transferFunction");
        syldet_layer_t &L = oc.layers[(size_t)i];
        L.inputs = (int32_t)inputs;
        L.outputs = (int32_t)outputs;
        if (tf == "TanSig") L.transfer = SYLDET_TF_TANSIG;
        else if (tf == "LogSig") L.transfer = SYLDET_TF_LOGSIG;
        else if (tf == "PureLin") L.transfer = SYLDET_TF_PURELIN;
        else if (tf == "SatLin") L.transfer = SYLDET_TF_SATLIN;
        else throw_invalid(nm + ".transferFunction");
        // NeuralNetLayer.init :340-342 (fatalError in the reference)
        if (inputs <= 0 || outputs <= 0)
            throw ParseFailure{SYLDET_ERR_LAYER_SHAPE, "Each layer must have at least one input and at least one output."};
    }

    const long long n_in = parse_int("processInputsCount", data);               // :262-266
    if (n_in < 0 || n_in > 4096) throw_invalid("processInputsCount");
    if (n_in > 0 && layer_count == 0)
        throw ParseFailure{SYLDET_ERR_LAYER_SHAPE, "Neural network must have 1 or more layers."};
    std::vector<ParsedFn> in_fns, out_fns;
    for (long long i = 0; i < n_in; i++)
        in_fns.push_back(parse_input_fn("processInputs" + std::to_string(i), oc.layers[0].inputs, data));
    const long long n_out = parse_int("processOutputsCount", data);             // :269-273
    if (n_out < 0 || n_out > 4096) throw_invalid("processOutputsCount");
    if (n_out > 0 && layer_count == 0)
        throw ParseFailure{SYLDET_ERR_LAYER_SHAPE, "Neural network must have 1 or more layers."};
    for (long long i = 0; i < n_out; i++)
        out_fns.push_back(parse_output_fn("processOutputs" + std::to_string(i), oc.layers[(size_t)layer_count - 1].outputs, data));
    if (layer_count == 0)                                                       // NeuralNet.init :244-246
        throw ParseFailure{SYLDET_ERR_LAYER_SHAPE, "Neural network must have 1 or more layers."};
    for (long long l = 1; l < layer_count; l++)                                 // NeuralNet.init :248-254
        if (oc.layers[l - 1].outputs != oc.layers[l].inputs)
            throw ParseFailure{SYLDET_ERR_LAYER_SHAPE, "Number of inputs for layer " + std::to_string(l) + " does not match previous outputs."};

    auto emplace_fns = [](std::vector<ParsedFn> &src, std::vector<syldet_fn_t> &fns,
                          std::vector<std::vector<float>> &xo, std::vector<std::vector<float>> &ga) {
        fns.resize(src.size());
        xo.resize(src.size());
        ga.resize(src.size());
        for (size_t i = 0; i < src.size(); i++) {
            fns[i].kind = src[i].kind;
            fns[i].count = (int32_t)src[i].xoff.size();
            fns[i].y = src[i].y;
            xo[i] = std::move(src[i].xoff);
            ga[i] = std::move(src[i].gain);
        }
    };
    emplace_fns(in_fns, oc.input_fns, oc.fn_xoff_in, oc.fn_gain_in);
    emplace_fns(out_fns, oc.output_fns, oc.fn_xoff_out, oc.fn_gain_out);

    // what SyllableDetector fixes on top of the file: hamming window
    // (SyllableDetector.swift:43), extractPower (:136), lastDetected on output 0 (:27-31)
    v.window = SYLDET_WINDOW_HAMMING;
    v.spectrum = SYLDET_SPECTRUM_POWER;
    v.rule = SYLDET_RULE_FIRST;
    oc.relink();
    return SYLDET_OK;
}

}  // namespace sd

// ---------------------------------------------------------------- C ABI (host-only part)

using namespace sd;

extern "C" {

int syldet_abi_version(void) { return SYLDET_ABI_VERSION; }

const char *syldet_last_error(void) { return g_last_error.c_str(); }

const char *syldet_strerror(int status)
{
    switch (status) {
    case SYLDET_OK: return "ok";
    case SYLDET_ERR_INVALID_ARGUMENT: return "invalid argument";
    case SYLDET_ERR_FFT_SIZE: return "The FFT size must be a power of 2 and at least the window length.";
    case SYLDET_ERR_OVERLAP: return "Invalid overlap value.";
    case SYLDET_ERR_FREQ_RANGE: return "The frequency range is invalid.";
    case SYLDET_ERR_INPUT_MISMATCH: return "The neural network's input count does not match the configuration settings.";
    case SYLDET_ERR_THRESHOLD_MISMATCH: return "The neural network's output count does not match the thresholds.";
    case SYLDET_ERR_LAYER_SHAPE: return "Invalid neural network layer shapes.";
    case SYLDET_ERR_BUFFER_FULL: return "Insufficient space on buffer.";
    case SYLDET_ERR_NO_DEVICE: return "No usable gfx950 HIP device.";
    case SYLDET_ERR_DEVICE: return "HIP runtime error.";
    case SYLDET_ERR_OUT_OF_MEMORY: return "Out of memory.";
    case SYLDET_ERR_PARSE_OPEN: return "unableToOpenPath";
    case SYLDET_ERR_PARSE_MISSING: return "missingValue";
    case SYLDET_ERR_PARSE_INVALID: return "invalidValue";
    case SYLDET_ERR_PARSE_LENGTH: return "mismatchedLength";
    case SYLDET_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
    }
}

int syldet_config_load_text(const char *path, syldet_config_t **out)
{
    if (!path || !out) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    std::unique_ptr<OwnedConfig> oc(new OwnedConfig());
    try {
        load_text(path, *oc);
    } catch (const ParseFailure &f) {
        return fail(f.status, f.text);
    } catch (const std::bad_alloc &) {
        return fail(SYLDET_ERR_OUT_OF_MEMORY, "out of memory");
    }
    Loaded *l = new Loaded();
    l->owned = oc.release();
    l->view = l->owned->view;
    *out = &l->view;
    return SYLDET_OK;
}

void syldet_config_free(syldet_config_t *cfg)
{
    if (!cfg) return;
    Loaded *l = reinterpret_cast<Loaded *>(cfg);
    delete l->owned;
    delete l;
}

int syldet_config_geometry(const syldet_config_t *cfg, syldet_geometry_t *out)
{
    if (!cfg || !out) return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL argument");
    if ((cfg->n_layers > 0 && !cfg->layers) || (cfg->n_input_fns > 0 && !cfg->input_fns) ||
        (cfg->n_output_fns > 0 && !cfg->output_fns))
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "NULL array in configuration");
    return compute_geometry(*cfg, out);
}

int syldet_frequency_index_range(int32_t fourier_length, double sampling_rate, double lo, double hi,
                                 int32_t *f0, int32_t *f1)
{
    if (!f0 || !f1 || fourier_length <= 0) return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    return frequency_index_range(fourier_length, sampling_rate, lo, hi, f0, f1);
}

int syldet_make_window(int32_t window, int32_t length, float *out)
{
    if (!out || length <= 0 || window < SYLDET_WINDOW_NONE || window > SYLDET_WINDOW_BLACKMAN)
        return fail(SYLDET_ERR_INVALID_ARGUMENT, "bad argument");
    make_window(window, length, out);
    return SYLDET_OK;
}

}  // extern "C"
