// wav.hpp -- minimal RIFF/WAVE reader for the command line tool: what AVAssetReader + the detector's
// audioSettings (SyllableDetector.swift:19-23: 32-bit float linear PCM) deliver in the reference, for
// the container this image can decode without AVFoundation.  PCM 8/16/24/32-bit and IEEE float
// 32/64-bit, plain or WAVE_FORMAT_EXTENSIBLE, any channel count.  Samples come back frame-major
// (interleaved) as fp32 in [-1, 1): integer PCM is divided by 2^(bits-1), like Core Audio's converter.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace wav {

struct Info {
    int format = 0;          // 1 PCM, 3 IEEE float
    int channels = 0;
    int bits = 0;
    double rate = 0.0;
    int64_t frames = 0;
    int64_t data_offset = 0; // byte offset of the sample data in the file
};

inline uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// Parses the header; on failure returns false with a reason in `err`.
inline bool probe(const std::string &path, Info &info, std::string &err)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open file"; return false; }
    unsigned char hdr[12];
    if (std::fread(hdr, 1, 12, f) != 12 || std::memcmp(hdr, "RIFF", 4) != 0 || std::memcmp(hdr + 8, "WAVE", 4) != 0) {
        std::fclose(f);
        err = "not a RIFF/WAVE file";
        return false;
    }
    bool have_fmt = false, have_data = false;
    int block_align = 0;
    int64_t pos = 12, data_bytes = 0;
    for (;;) {
        unsigned char ch[8];
        if (std::fseek(f, (long)pos, SEEK_SET) != 0 || std::fread(ch, 1, 8, f) != 8) break;
        const uint32_t size = rd32(ch + 4);
        if (std::memcmp(ch, "fmt ", 4) == 0) {
            unsigned char b[40] = {0};
            const size_t want = size < 40 ? size : 40;
            if (size < 16 || std::fread(b, 1, want, f) != want) { err = "truncated fmt chunk"; break; }
            info.format = rd16(b);
            info.channels = rd16(b + 2);
            info.rate = (double)rd32(b + 4);
            block_align = rd16(b + 12);
            info.bits = rd16(b + 14);
            if (info.format == 0xFFFE && size >= 26) info.format = rd16(b + 24);   // WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
            have_fmt = true;
        } else if (std::memcmp(ch, "data", 4) == 0) {
            info.data_offset = pos + 8;
            data_bytes = size;
            have_data = true;
            break;                                     // sample data is the last thing we need
        }
        pos += 8 + (int64_t)size + (size & 1);         // chunks are word aligned
    }
    if (have_fmt && have_data) {
        // a streamed file may carry 0 or 0xFFFFFFFF as its data size: trust the file length then
        std::fseek(f, 0, SEEK_END);
        const int64_t end = (int64_t)std::ftell(f);
        if (data_bytes == 0 || data_bytes == 0xFFFFFFFFll || info.data_offset + data_bytes > end) data_bytes = end - info.data_offset;
    }
    std::fclose(f);
    if (!have_fmt) { if (err.empty()) err = "no fmt chunk"; return false; }
    if (!have_data) { err = "no data chunk"; return false; }
    if (info.channels <= 0) { err = "no audio channels"; return false; }
    const bool pcm = info.format == 1 && (info.bits == 8 || info.bits == 16 || info.bits == 24 || info.bits == 32);
    const bool flt = info.format == 3 && (info.bits == 32 || info.bits == 64);
    if (!pcm && !flt) { err = "unsupported sample format (format tag " + std::to_string(info.format) + ", " + std::to_string(info.bits) + " bits)"; return false; }
    if (block_align != info.channels * info.bits / 8) { err = "inconsistent block alignment"; return false; }
    info.frames = data_bytes / block_align;
    return true;
}

// Reads every frame as interleaved fp32.
inline bool read(const std::string &path, Info &info, std::vector<float> &out, std::string &err)
{
    if (!probe(path, info, err)) return false;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open file"; return false; }
    const size_t n = (size_t)info.frames * (size_t)info.channels, bps = (size_t)info.bits / 8;
    std::vector<unsigned char> raw(n * bps);
    std::fseek(f, (long)info.data_offset, SEEK_SET);
    const size_t got = std::fread(raw.data(), 1, raw.size(), f);
    std::fclose(f);
    if (got != raw.size()) { err = "truncated sample data"; return false; }
    out.resize(n);
    const unsigned char *p = raw.data();
    if (info.format == 3 && info.bits == 32) {
        std::memcpy(out.data(), p, n * 4);
    } else if (info.format == 3) {
        for (size_t i = 0; i < n; i++) { double d; std::memcpy(&d, p + 8 * i, 8); out[i] = (float)d; }
    } else if (info.bits == 16) {
        for (size_t i = 0; i < n; i++) out[i] = (float)(int16_t)rd16(p + 2 * i) * (1.0f / 32768.0f);
    } else if (info.bits == 8) {
        for (size_t i = 0; i < n; i++) out[i] = (float)((int)p[i] - 128) * (1.0f / 128.0f);
    } else if (info.bits == 24) {
        for (size_t i = 0; i < n; i++) {
            const int32_t v = (int32_t)((uint32_t)p[3 * i] << 8 | (uint32_t)p[3 * i + 1] << 16 | (uint32_t)p[3 * i + 2] << 24) >> 8;
            out[i] = (float)v * (1.0f / 8388608.0f);
        }
    } else {
        for (size_t i = 0; i < n; i++) out[i] = (float)((double)(int32_t)rd32(p + 4 * i) * (1.0 / 2147483648.0));
    }
    return true;
}

}  // namespace wav
