// fused_plan.hpp -- host-side tables of the fused engine (see fused_plan.cpp).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "syldet_internal.hpp"

namespace sd {

struct FusedPlan {
    bool ok = false;
    std::string reason;              // why the configuration does not fit, when !ok
    FusedDesc desc{};                // device pointers are filled in by the owner after upload
    std::vector<uint16_t> dfrag;     // f16 bit patterns
    std::vector<uint16_t> afrag, afrag_t, afrag_w;
    std::vector<uint16_t> sfrag;     // the folded basis of the symmetric-fold kernel
    std::vector<float> slone;
    std::vector<uint16_t> sfrag2, afrag_t2, afrag_w2;   // the second-fold instantiation's basis and first layer
    std::vector<float> swin2, s2c;
    std::vector<int> koff;
    std::vector<float> bias0, rvec, w1, b1, out_params;
};

struct MlpxPlan {
    bool ok = false;
    std::string reason;
    MlpxDesc desc{};
    std::vector<uint16_t> afrag;
    std::vector<float> bias0, w1;
};
struct BdftPlan {
    bool ok = false;
    BdftDesc desc{};
    MlpxDesc md{};                   // the network stage's descriptor with 128-bin columns (filled in by the owner after upload)
    std::vector<uint16_t> basis, afrag;
    std::vector<float> cre;
};
// frames of four hops on the block-transform kernel (kernels_bdft.hip), when the configuration is of the matrix-core network
// stage's class, W = N = 4 hop with hop 128 or 256, and the window is rectangular, Hann or Hamming
bool make_bdft_plan(const syldet_config_t &cfg, const syldet_geometry_t &geom, const MlpxPlan &mlpx, BdftPlan &plan);
// the matrix-core network stage of the generic engine (kernels_mlpx.hip), when the configuration is of its class
bool make_mlpx_plan(const syldet_config_t &cfg, const syldet_geometry_t &geom, MlpxPlan &plan);

bool make_fused_plan(const syldet_config_t &cfg, const syldet_geometry_t &geom, FusedPlan &plan);
// choose passes per workgroup for a batch of E evaluations x C channels
void fused_segmentation(FusedDesc &d, int64_t E, int C);

}  // namespace sd
