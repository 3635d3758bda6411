// syllable-detector-cli -- the reference's command line tool (SyllableDetectorCLI/main.swift:19-131,
// TrackDetector.swift:45-105) over libsyldet: same options, same output lines
//     channel,sample,seconds,out0[,out1...]
// one per detection event (any output at or above its threshold, TrackDetector.swift:72-77; debounce
// :80,:99), a line with the file name first when more than one file is given (main.swift:122-124).
// All tracks of a file are one batch on the GPU: decode -> H2D -> de-interleave [-> rate conversion when the
// file's rate differs from the network's] -> fused STFT + network kernel -> flags/outputs -> host.
//
// Differences a user can see: the reference decodes anything AVFoundation can, this tool reads WAV; the
// reference has Core Audio deliver the network's rate (SyllableDetector.swift:19-23), this tool converts the decoded
// file by linear interpolation with fp64 positions (syldet_convert_rate_device); events of different channels are interleaved buffer by buffer like the
// reference's read loop (main.swift:126-130), with --chunk frames per buffer (AVAssetReader's buffer size is
// not specified; 8192 is what it typically vends for linear PCM).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "syldet.h"
#include "wav.hpp"

namespace {

constexpr int kExUsage = 64;   // EX_USAGE, main.swift:40

void usage(FILE *to)
{
    std::fprintf(to,
                 "Usage: syllable-detector-cli -n <net> [-a <audio>]... [-d <seconds>] [--device <k>] [--chunk <frames>] [--format <shortest|swift4>] [--probe]\n"
                 "  -n, --net <net>:\n      Path to trained network file.\n"
                 "  -a, --audio <audio>:\n      Path to the audio file to process.\n"
                 "  -d, --debounce <seconds>:\n      Number of seconds to debounce triggers.\n"
                 "      --device <k>:\n      HIP device to run on (default 0).\n"
                 "      --chunk <frames>:\n      Frames per decode buffer when interleaving events of several channels (default 8192; 0: channel by channel).\n"
                 "      --format <shortest|swift4>:\n      How numbers are printed: the shortest digits that round-trip (Swift 4.2 and later; default) or 15 / 6 significant digits (Swift 4.0, the toolchain the project declares: the example line below).\n"
                 "      --probe:\n      Only print what the audio files contain; does not touch the GPU.\n"
                 "The command line will write a comma-separated list of detection events (when the network has at least one output above threshold) to standard out. For example, it might output:\n"
                 "\n\t0,1593298,36.1292063492063,0.918557\n\n"
                 "The columns are:\n"
                 "1. The track or channel number from the audio file (starting with 0).\n"
                 "2. The sample number from the audio when detection occurred.\n"
                 "3. The timestamp from the audio when detection occurred.\n"
                 "4. The first neural network output. Note that there may be additional columns for additional outputs.\n");
}

// Swift's description of a Double / Float: the shortest digits that round-trip, with ".0" for whole numbers.
template <typename F>
std::string swift_number(F v)
{
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v);
    std::string s(buf, r.ptr);
    if (s.find_first_of(".en") == std::string::npos) s += ".0";   // "2" -> "2.0"; leaves "1e+16", "inf", "nan"
    return s;
}

// The same under Swift 4.0 / 4.1, the toolchain the project declares (SWIFT_VERSION = 4.0, project.pbxproj:593): before
// Swift 4.2 `description` printed "%0.*g" with digits10 significant digits -- 15 for Double, 6 for Float -- and appended ".0" when
// the text held neither '.', 'e' nor a letter.  The one output line the reference holds, in its help text (main.swift:33),
// "0,1593298,36.1292063492063,0.918557", is in this form: 15 and 6 digits.
template <typename F>
std::string swift4_number(F v)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%0.*g", sizeof(F) == 8 ? 15 : 6, (double)v);
    std::string s(buf);
    if (s.find_first_of(".eEn") == std::string::npos) s += ".0";          // ("inf", "nan" keep their letters)
    return s;
}

bool g_swift4 = false;                                     // --format swift4
template <typename F>
std::string number(F v) { return g_swift4 ? swift4_number(v) : swift_number(v); }

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess; }
};

struct Event { int64_t buffer; int channel; int64_t sample; int64_t eval; };

int process_file(const std::string &path, const syldet_config_t *cfg, int device, double debounce_s, bool have_debounce, int64_t chunk)
{
    wav::Info info;
    std::vector<float> frames;
    std::string err;
    if (!wav::read(path, info, frames, err)) {
        std::fprintf(stderr, "Unable to read %s: %s\n", path.c_str(), err.c_str());
        return 1;
    }
    const int C = info.channels;
    if (C <= 0 || info.frames <= 0) {
        std::fprintf(stderr, "No audio tracks found in %s.\n", path.c_str());
        return 1;
    }
    syldet_t *h = nullptr;
    if (int st = syldet_create(cfg, C, device, SYLDET_ENGINE_AUTO, &h)) {
        std::fprintf(stderr, "Unable to create the detector: %s: %s\n", syldet_strerror(st), syldet_last_error());
        return 2;
    }
    syldet_geometry_t g;
    syldet_get_geometry(h, &g);
    const int n_out = g.outputs;
    const bool resample = info.rate != cfg->sampling_rate;
    int rc = 0;
    hipStream_t stream = nullptr;
    std::vector<float> out;
    std::vector<uint8_t> flags;
    int64_t E = 0;
    do {
        if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&stream) != hipSuccess) { rc = 2; break; }
        const int64_t n = info.frames;
        DevBuf d_inter, d_planar, d_res, d_out, d_flags;
        if (!d_inter.alloc((size_t)n * C * sizeof(float))) { rc = 2; break; }
        if (hipMemcpyAsync(d_inter.p, frames.data(), (size_t)n * C * sizeof(float), hipMemcpyHostToDevice, stream) != hipSuccess) { rc = 2; break; }
        int64_t S = n, res_stride = 0;
        int st = 0;
        if (resample) {
            // The reference's tool has AVFoundation deliver every track at the network's rate (SyllableDetector.swift:19-23,
            // TrackDetector.swift:35); here: linear interpolation of the whole decoded file with fp64 positions
            // (syldet_convert_rate_device; ResamplerLinear is the live path's streaming object, not a file converter)
            res_stride = syldet_convert_rate_count(n, info.rate, cfg->sampling_rate);
            if (!d_planar.alloc((size_t)C * n * sizeof(float)) || !d_res.alloc((size_t)C * (res_stride > 0 ? res_stride : 1) * sizeof(float))) { rc = 2; break; }
            st = syldet_deinterleave_device((const float *)d_inter.p, n, C, 0, C, (float *)d_planar.p, n, stream);
            if (!st) st = syldet_convert_rate_device((const float *)d_planar.p, n, n, C, info.rate, cfg->sampling_rate, (float *)d_res.p, res_stride, &S, stream);
            if (st) {
                std::fprintf(stderr, "Unable to process %s: %s: %s\n", path.c_str(), syldet_strerror(st), syldet_last_error());
                rc = 2;
                break;
            }
        }
        E = syldet_count_evals(h, S);
        if (E <= 0) break;                                  // shorter than one evaluation: no events
        if (!d_out.alloc((size_t)C * E * n_out * sizeof(float)) || !d_flags.alloc((size_t)C * E)) { rc = 2; break; }
        if (!resample) st = syldet_run_interleaved_device(h, (const float *)d_inter.p, n, C, (float *)d_out.p, (uint8_t *)d_flags.p, stream);
        else st = syldet_run_device(h, (const float *)d_res.p, S, res_stride, (float *)d_out.p, (uint8_t *)d_flags.p, stream);
        if (st) {
            std::fprintf(stderr, "Unable to process %s: %s: %s\n", path.c_str(), syldet_strerror(st), syldet_last_error());
            rc = 2;
            break;
        }
        out.resize((size_t)C * E * n_out);
        flags.resize((size_t)C * E);
        if (hipMemcpyAsync(out.data(), d_out.p, out.size() * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipMemcpyAsync(flags.data(), d_flags.p, flags.size(), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) { rc = 2; break; }
    } while (false);
    if (rc == 2 && hipPeekAtLastError() != hipSuccess) std::fprintf(stderr, "Unable to process %s: %s\n", path.c_str(), hipGetErrorString(hipGetLastError()));
    if (stream) (void)hipStreamDestroy(stream);
    syldet_destroy(h);
    if (rc || E <= 0) return rc;

    // events: sample number of evaluation e = first_index + e*hop (TrackDetector.swift:39-43,67-68); debounce :80,:99
    const int64_t debounce_frames = have_debounce ? (int64_t)(debounce_s * cfg->sampling_rate) : 0;   // Int(newValue * samplingRate), :24
    std::vector<Event> events;
    for (int c = 0; c < C; c++) {
        int64_t until = -1;
        for (int64_t e = 0; e < E; e++) {
            if (!flags[(size_t)c * E + e]) continue;
            const int64_t idx = (int64_t)g.first_index + e * (int64_t)g.hop;
            if (!(until < idx)) continue;
            until = idx + debounce_frames;
            events.push_back({chunk > 0 ? (idx - 1) / chunk : 0, c, idx, e});
        }
    }
    // the reference's loop hands every track one buffer per round (main.swift:126-130)
    std::stable_sort(events.begin(), events.end(), [](const Event &a, const Event &b) {
        if (a.buffer != b.buffer) return a.buffer < b.buffer;
        if (a.channel != b.channel) return a.channel < b.channel;
        return a.sample < b.sample;
    });
    for (const Event &ev : events) {
        std::string line = std::to_string(ev.channel) + "," + std::to_string(ev.sample) + "," +
                           number((double)ev.sample / cfg->sampling_rate);
        for (int o = 0; o < n_out; o++) line += "," + number(out[((size_t)ev.channel * E + ev.eval) * n_out + o]);
        std::puts(line.c_str());
    }
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    std::string net;
    std::vector<std::string> audio;
    double debounce = 0.0;
    bool have_debounce = false, probe = false;
    int device = 0;
    int64_t chunk = 8192;
    auto value = [&](int &i, const char *name) -> const char * {
        if (i + 1 >= argc) {
            std::fprintf(stderr, "Missing value for %s.\n", name);
            usage(stdout);
            std::exit(kExUsage);
        }
        return argv[++i];
    };
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-n" || a == "--net") net = value(i, "--net");
        else if (a == "-a" || a == "--audio") audio.push_back(value(i, "--audio"));
        else if (a == "-d" || a == "--debounce") {
            const char *v = value(i, "--debounce");
            char *end = nullptr;
            debounce = std::strtod(v, &end);
            have_debounce = end && *end == 0 && end != v;   // Double.init(String): nil (no debounce) when not a number
        } else if (a == "--device") device = std::atoi(value(i, "--device"));
        else if (a == "--chunk") chunk = std::atoll(value(i, "--chunk"));
        else if (a == "--probe") probe = true;
        else if (a == "--format") {
            const std::string f = value(i, "--format");
            if (f != "shortest" && f != "swift4") { usage(stdout); return kExUsage; }
            g_swift4 = f == "swift4";
        } else if (a == "--format-line") {
            // formats one event line from its parts (channel sample rate out0 [out1 ...]; the outputs as fp32) exactly as the
            // event loop does: the known-answer test of the number formats, no audio and no GPU involved
            if (i + 3 >= argc) { usage(stdout); return kExUsage; }
            const long long smp = std::atoll(argv[i + 2]);
            std::string line = std::string(argv[i + 1]) + "," + std::to_string(smp) + "," + number((double)smp / std::strtod(argv[i + 3], nullptr));
            for (int k = i + 4; k < argc; k++) line += "," + number(std::strtof(argv[k], nullptr));
            std::puts(line.c_str());
            return 0;
        } else {                                              // -h, --help and anything unknown: usage text, EX_USAGE (main.swift:27-41)
            usage(stdout);
            return kExUsage;
        }
    }
    if (probe) {
        int bad = 0;
        for (const std::string &p : audio) {
            wav::Info info;
            std::string err;
            if (!wav::probe(p, info, err)) { std::fprintf(stderr, "Unable to read %s: %s\n", p.c_str(), err.c_str()); bad = 1; continue; }
            std::printf("%s: %d channel(s), %s Hz, %s %d-bit, %lld frames\n", p.c_str(), info.channels, swift_number(info.rate).c_str(),
                        info.format == 3 ? "float" : "pcm", info.bits, (long long)info.frames);
        }
        return bad;
    }
    if (net.empty()) {                                      // the option is .required() in the reference (main.swift:21)
        usage(stdout);
        return kExUsage;
    }
    syldet_config_t *cfg = nullptr;
    if (int st = syldet_config_load_text(net.c_str(), &cfg)) {
        std::fprintf(stderr, "Unable to load the network configuration: %s: %s\n", syldet_strerror(st), syldet_last_error());
        return 1;
    }
    cfg->rule = SYLDET_RULE_ANY;                            // any output above its threshold, TrackDetector.swift:72-77
    int rc = 0;
    for (const std::string &p : audio) {
        if (audio.size() > 1) std::printf("%s\n", p.c_str());   // main.swift:122-124
        std::fflush(stdout);
        const int r = process_file(p, cfg, device, debounce, have_debounce, chunk);
        if (r == 2) rc = 2;                                 // device trouble is fatal for the exit code; an unreadable file is skipped
    }
    syldet_config_free(cfg);
    return rc;
}
