// syldet_internal.hpp -- shared declarations of libsyldet's host side (not installed).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "syldet.h"

namespace sd {

// Thread-local text of the last failing call (syldet_last_error).
void set_error(const std::string &msg);
int fail(int status, const std::string &msg);

// Deep, self-owned copy of a syldet_config_t.  `view` points into the vectors.
struct OwnedConfig {
    syldet_config_t view{};
    std::vector<syldet_fn_t> input_fns, output_fns;
    std::vector<syldet_layer_t> layers;
    std::vector<std::vector<float>> fn_xoff_in, fn_gain_in, fn_xoff_out, fn_gain_out;
    std::vector<std::vector<float>> weights, biases;
    std::vector<double> thresholds;

    OwnedConfig() = default;
    OwnedConfig(const OwnedConfig &) = delete;
    OwnedConfig &operator=(const OwnedConfig &) = delete;

    int assign(const syldet_config_t &src);   // validates pointers, copies arrays
    void relink();                             // re-point view at the vectors
};

// SyllableDetector.init's validation (SyllableDetector.swift:42-60) plus the STFT
// constructor's (CircularShortTimeFourierTransform.swift:61-96).
int compute_geometry(const syldet_config_t &cfg, syldet_geometry_t *out);

void make_window(int window, int length, float *out);

}  // namespace sd
