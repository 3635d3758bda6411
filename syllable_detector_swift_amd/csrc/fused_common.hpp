// fused_common.hpp -- device helpers shared by the fused engine kernels (kernels_fused.hip, kernels_fused_r.hip).
// gfx950 only.  wave = 64.
#pragma once

#include "kernels.hpp"

namespace sd {
namespace fused_dev {

// layout of the constant block in LDS (floats)
constexpr int kCstThr = 0 /* 16 doubles */, kCstOut = 32;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned int uint32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int uint32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ half8 as_half8(uint32x4 v)
{
    union { uint32x4 u; half8 h; } c;
    c.u = v;
    return c.h;
}
__device__ __forceinline__ floatx4 as_floatx4(uint32x4 v)
{
    union { uint32x4 u; floatx4 f; } c;
    c.u = v;
    return c.f;
}

__device__ __forceinline__ float pow2f(int e)   // 2^e for e in [-126, 127]
{
    return __uint_as_float((unsigned)(e + 127) << 23);
}

// max(m, |x|, |y|) in one instruction (fmaxf's NaN canonicalisation costs an extra op per value)
__device__ __forceinline__ float absmax3(float m, float x, float y)
{
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(x), "v"(y), "v"(m));
    return r;
}

// Transfer functions (NeuralNet.swift:185-228).  tanh/logistic through the hardware exp2/rcp, written so
// that NaN and the infinities fall out of the arithmetic itself (no selects, no branches: silence gives 0/0 in
// l2normalize and the reference then never detects): absolute error below 4e-7, far inside the 1e-5 bar.
__device__ __forceinline__ float transfer_fn(int tf, float x)
{
    if (tf == 0)                                     // TanSig: 1 - 2 / (e^{2x} + 1); e^{2x} = inf gives 1, 0 gives -1
        return fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.885390081777927f) + 1.0f), 1.0f);
    if (tf == 1)                                     // LogSig: 1 / (1 + e^{-x})
        return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * -1.4426950408889634f) + 1.0f);
    if (tf == 3) return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);   // SatLin (NaN falls through both tests)
    return x;                                        // PureLin
}

// The buffer descriptor of one pass's samples ends one past the last sample any existing frame
// reads, so quads beyond it come back as zeros from the hardware bounds check: no per-lane guards.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float *row, int64_t first, int64_t s_eff, int nsmp)
{
    int64_t left = s_eff - first;
    left = left < 0 ? 0 : (left > nsmp ? nsmp : left);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(row + first), 0, (int)left * 4, 0x00020000);
}

// 8 consecutive halves from an 8-byte aligned LDS address (two ds_read_b64)
[[maybe_unused]] __device__ __forceinline__ half8 lds_half8(const _Float16 *p)
{
    const uint32x2 lo = *reinterpret_cast<const uint32x2 *>(p), hi = *reinterpret_cast<const uint32x2 *>(p + 4);
    uint32x4 u = {lo[0], lo[1], hi[0], hi[1]};
    return as_half8(u);
}

// LDS byte address of a pointer into shared memory (the low half of its flat address)
[[maybe_unused]] __device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(uintptr_t)p; }

__device__ __forceinline__ floatx4 mfma(half8 a, half8 b, floatx4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// x*sx -> f16 hi (round to nearest) and f16 lo = the exact remainder x*sx - hi rounded to f16, for two
// values at once (packed words): hi + lo == x*sx to 2^-22 relative.  sx is a power of two at every call site, so x*sx
// is exact; hi = RNE(x*sx); v_fma_mix_f32 gives x*sx - hi exactly in fp32 (mixed-width sources: hi is read as a half);
// lo = RNE of it.  Six one-slot instructions a pair: v_fma_mixlo/hi_f16, which would do it in four, take two issue slots
// each and a wait state between the halves of a register (tools/ubench/valu_rates, tick_costs).
__device__ __forceinline__ void split_pair_scaled(float a, float b, float sx, unsigned &hi, unsigned &lo)
{
    unsigned h, l;
    float ta, tb, ra, rb;
    asm("v_mul_f32 %0, %1, %2" : "=v"(ta) : "v"(a), "v"(sx));
    asm("v_mul_f32 %0, %1, %2" : "=v"(tb) : "v"(b), "v"(sx));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(ta), "v"(tb));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "v"(sx), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "v"(sx), "v"(h));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(ra), "v"(rb));
    hi = h;
    lo = l;
}

// Cross-lane helpers without LDS round trips (a ds_bpermute costs an LDS latency on the critical path each):
// gfx950's v_permlane16_swap / v_permlane32_swap exchange 16-lane rows / 32-lane halves between two registers.
// With both operands = x the two results are x and its xor-16 (xor-32) partner, in some order per lane.
__device__ __forceinline__ float xor16_sum(float x)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float x)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// maximum over the wave of a non-negative float (compared as bit patterns), wave-uniform result
__device__ __forceinline__ float wave_max_nonneg(float x)
{
    unsigned u = __float_as_uint(x);
    u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x141, 0xF, 0xF, false));   // row_half_mirror
    u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x140, 0xF, 0xF, false));   // row_mirror
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    u = max(r[0], r[1]);
    r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    u = max(r[0], r[1]);
    return __uint_as_float(__builtin_amdgcn_readfirstlane(u));
}

}  // namespace fused_dev
}  // namespace sd
