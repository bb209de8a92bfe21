// wave_ops.h -- wave64 all-reduces on the VALU only (gfx950): four DPP steps inside each row of 16 lanes, then
// v_permlane16_swap / v_permlane32_swap (new in gfx950) across the rows -- 8 instructions, against six ds_bpermute round
// trips through the LDS crossbar for the __shfl_xor butterfly.  Every lane ends with the same value (each step combines a
// lane with its mirror partner, and a + b == b + a bit for bit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace idl_dev {

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1;          // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm:[2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane i <-> 7 - i inside each 8
constexpr int DPP_MIRROR = 0x140;       // lane i <-> 15 - i inside each row

// value of the lane 16 (32) positions away, as a pair with this lane's own: r[0], r[1] hold {own, partner} in some order
__device__ __forceinline__ float add_xor16(float v)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float add_xor32(float v)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float max_xor16(float v)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float max_xor32(float v)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ float row_sum16(float v)      // sum over the 16 lanes of this lane's row
{
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    v += dpp_f<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) { return add_xor32(add_xor16(row_sum16(v))); }

// the same all-reduce for float64: the two 32-bit halves travel separately through the same DPP / permlane-swap steps
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double add_xor16_d(double v)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double add_xor32_d(double v)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double wave_sum_d(double v)
{
    v += dpp_d<DPP_XOR1>(v);
    v += dpp_d<DPP_XOR2>(v);
    v += dpp_d<DPP_HALF_MIRROR>(v);
    v += dpp_d<DPP_MIRROR>(v);
    return add_xor32_d(add_xor16_d(v));
}
// smallest value of the wave, float64 and unsigned 64-bit (a packed (number, position) pair), in every lane
__device__ __forceinline__ double min_xor16_d(double v)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    return fmin(__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1]));
}
__device__ __forceinline__ double min_xor32_d(double v)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    return fmin(__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1]));
}
__device__ __forceinline__ double wave_min_d(double v)
{
    v = fmin(v, dpp_d<DPP_XOR1>(v));
    v = fmin(v, dpp_d<DPP_XOR2>(v));
    v = fmin(v, dpp_d<DPP_HALF_MIRROR>(v));
    v = fmin(v, dpp_d<DPP_MIRROR>(v));
    return min_xor32_d(min_xor16_d(v));
}
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, true);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
    v = umin64(v, dpp_u64<DPP_XOR1>(v));
    v = umin64(v, dpp_u64<DPP_XOR2>(v));
    v = umin64(v, dpp_u64<DPP_HALF_MIRROR>(v));
    v = umin64(v, dpp_u64<DPP_MIRROR>(v));
    {
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(v >> 32), (unsigned)(v >> 32), false, false);
        v = umin64(((uint64_t)hi[0] << 32) | lo[0], ((uint64_t)hi[1] << 32) | lo[1]);
    }
    {
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(v >> 32), (unsigned)(v >> 32), false, false);
        v = umin64(((uint64_t)hi[0] << 32) | lo[0], ((uint64_t)hi[1] << 32) | lo[1]);
    }
    return v;
}

__device__ __forceinline__ float wave_max_f(float v)
{
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
    return max_xor32(max_xor16(v));
}

// acc + t * t as a product and a sum, each rounded (sklearn's distance loops, compiled without fused multiply-add).  hipcc contracts
// x * y + z into one fma by default and __dmul_rn / __dadd_rn are plain operators there, so the contraction is switched off here.
__device__ __forceinline__ double square_then_add(double acc, double t)
{
#pragma clang fp contract(off)
    const double p = t * t;
    return acc + p;
}

}  // namespace idl_dev
