// planes.h -- the two-fp16-plane form of an fp32 tensor (device helpers shared by planes.hip, scaler_device.h, wgrad_device.h):
//     v 2^k = v0 + v1,   v0 = fp16(v 2^k) (round to nearest even),   v1 = fp16(v 2^k - v0)
// -- 22 significand bits in 4 bytes, as many bytes as the fp32 value.  A product of two such tensors on the fp16 matrix cores
// (v_mfma_f32_32x32x16_f16, fp32 accumulators) as x0 y0 + x0 y1 + x1 y0 drops x1 y1, 2^-22 of a product; measured against a float64
// product it is closer than an fp32 GEMM, which rounds after every one of its additions (DESIGN.md History, profiles/r05_probe_split_mfma.txt).
//
// The scales are FIXED powers of two, so a producer never has to know a tensor's largest entry before it writes:
//   x  (a standardised feature, |x| <= sqrt(N - 1)):   2^3   -- in range up to N = 6.7e7 sequences (the host refuses beyond)
//   W1 (Linear(F,512).weight, U(+-1/sqrt(F)) at start): 2^12  -- in range up to |w| = 15.8; an entry beyond that is clamped AND raises the
//                                                               overflow flag the host reads at the end of an epoch
// An entry 2^-15 of the range still has its absolute error below 2^-25 2^-k (fp16's subnormal spacing): 2^-28 of a typical entry.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace idl_planes {

constexpr int X_EXP = 3, W_EXP = 12;
// dr1 (the gradient at the layer-1 activations: its range moves with the training) carries 2^k with k chosen so that the largest |dr1| of the
// PREVIOUS step sits in [2^8, 2^9): 128 x headroom below fp16's largest value, an entry 2^-15 of the largest still exact to 2^-33 of it.  Its
// words (int32, one buffer per voter): [0] the exponent the step's planes carry (mid_bwd writes, the dW1 tiles read), [1] the exponent of the next
// step (the dW1 launch writes it from the maxima; DR1_K_FIRST when a voter begins), [4 .. 4 + 63] the largest |dr1| of mid_bwd's workgroups as bits.
constexpr int DR1_K_FIRST = 10, DR1_K_TARGET = 9, DR1_WORDS = 4 + 64;
constexpr float LIMIT = 65000.f;       // (fp16's largest finite value is 65 504)

__device__ __forceinline__ uint32_t pack2(_Float16 a, _Float16 b)
{
    return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
}

// four scaled values -> 8 bytes of the high plane, 8 of the low; returns whether any left the range (then clamped)
__device__ __forceinline__ bool split4(float s0, float s1, float s2, float s3, uint2 &hi, uint2 &lo)
{
    // (NOT "max > LIMIT": fmaxf drops a NaN and fmed3 maps it to a bound -- a diverged tensor would give finite planes and no flag)
    const bool over = !(fabsf(s0) <= LIMIT) || !(fabsf(s1) <= LIMIT) || !(fabsf(s2) <= LIMIT) || !(fabsf(s3) <= LIMIT);
    s0 = __builtin_amdgcn_fmed3f(s0, -LIMIT, LIMIT); s1 = __builtin_amdgcn_fmed3f(s1, -LIMIT, LIMIT);
    s2 = __builtin_amdgcn_fmed3f(s2, -LIMIT, LIMIT); s3 = __builtin_amdgcn_fmed3f(s3, -LIMIT, LIMIT);
    const _Float16 h0 = (_Float16)s0, h1 = (_Float16)s1, h2 = (_Float16)s2, h3 = (_Float16)s3;
    const _Float16 l0 = (_Float16)(s0 - (float)h0), l1 = (_Float16)(s1 - (float)h1), l2 = (_Float16)(s2 - (float)h2), l3 = (_Float16)(s3 - (float)h3);
    hi = uint2{pack2(h0, h1), pack2(h2, h3)};
    lo = uint2{pack2(l0, l1), pack2(l2, l3)};
    return over;
}

}  // namespace idl_planes
