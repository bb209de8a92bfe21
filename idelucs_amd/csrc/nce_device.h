// nce_device.h -- the InfoNCE tile product shared by nce_fused.hip and the fused InfoNCE-pass-2 + middle-backward kernel of
// train_step.hip (see nce_fused.hip for the tiling).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nce_dev {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// S'[j][r] tile: rows j = j0..j0+15 (A operand), columns r = r0..r0+15 (B operand, preloaded in rb[])
__device__ __forceinline__ f32x4 sim_tile(const float *f, int j0, const float (&rb)[16], int l, int q)
{
    const float4 *src = (const float4 *)(f + (int64_t)(j0 + l) * 64 + 16 * q);
    const float4 a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
    const float a[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], rb[ks], acc, 0, 0, 0);
    return acc;
}

// two tiles at once (rows ja.. and jb..): both tiles' loads are in flight before the first product and the two dependent MFMA
// chains (40 cycles of latency per link) interleave
__device__ __forceinline__ void sim_tile2(const float *f, int ja, int jb, const float (&rb)[16], int l, int q, f32x4 &sa, f32x4 &sb)
{
    const float4 *pa = (const float4 *)(f + (int64_t)(ja + l) * 64 + 16 * q), *pb = (const float4 *)(f + (int64_t)(jb + l) * 64 + 16 * q);
    const float4 a0 = pa[0], a1 = pa[1], a2 = pa[2], a3 = pa[3], b0 = pb[0], b1 = pb[1], b2 = pb[2], b3 = pb[3];
    const float a[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
    const float b[16] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
    sa = f32x4{0.f, 0.f, 0.f, 0.f}; sb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        sa = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], rb[ks], sa, 0, 0, 0);
        sb = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ks], rb[ks], sb, 0, 0, 0);
    }
}

__device__ __forceinline__ void load_rows(const float *f, int r0, int l, int q, float (&rb)[16])
{
    const float4 *src = (const float4 *)(f + (int64_t)(r0 + l) * 64 + 16 * q);
    const float4 b0 = src[0], b1 = src[1], b2 = src[2], b3 = src[3];
    rb[0] = b0.x; rb[1] = b0.y; rb[2] = b0.z; rb[3] = b0.w; rb[4] = b1.x; rb[5] = b1.y; rb[6] = b1.z; rb[7] = b1.w;
    rb[8] = b2.x; rb[9] = b2.y; rb[10] = b2.z; rb[11] = b2.w; rb[12] = b3.x; rb[13] = b3.y; rb[14] = b3.z; rb[15] = b3.w;
}

}  // namespace nce_dev
