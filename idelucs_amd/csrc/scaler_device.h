// scaler_device.h -- device code shared by scaler.hip (idl_gather_pairs*) and train_step.hip (the optimizer launch that
// also assembles the NEXT batch): StandardScaler.transform arithmetic and the row gather.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "planes.h"

namespace idl_dev {

// float32 in -> float32 out: both ops in float64, rounded to float32 after each (numpy in-place ufunc on a float32 array
// with a float64 operand; reference idelucs/utils.py:361-366).
__device__ __forceinline__ float std_f32(float x, double m, double s)
{
    const float t = (float)((double)x - m);
    return (float)((double)t / s);
}

// same result with a precomputed r = RN(1/s): q = t r, q' = fma(fma(-q, s, t), r, q) is the correctly rounded float64
// quotient (Markstein; 3.8e9 random (t, s) pairs checked against IEEE division, tools/markstein_check_f64.c)
__device__ __forceinline__ float std_f32_rcp(float x, double m, double s, double r)
{
    const double t = (double)(float)((double)x - m);
    const double q = t * r;
    return (float)fma(fma(-q, s, t), r, q);
}

struct GatherArgs {
    const float *feats;
    int64_t n, f, view_stride;
    const int64_t *pair_idx;
    const int64_t *base;        // device int64 offset into pair_idx (may be NULL = 0)
    int64_t batch, n_pairs;     // rows whose pair index falls at or beyond n_pairs are skipped (n_pairs < 0: no limit)
    const double *mean, *scale, *inv_scale;
    float *y;
    int64_t base_add;           // added to *base (a launch that assembles the batch AFTER the one the offset points at)
    uint16_t *yh, *yl;          // optional (gather_tile only): the batch ALSO as two fp16 planes [2 batch][f] (planes.h, scale 2^X_EXP)
    int *over;                  // ... and the flag raised when an entry left the planes' range (|x| > 8 125: clamped in the planes)
};

// one workgroup copies + standardises one output row; rows [0, batch) are the "true" halves, [batch, 2*batch) the "modified" ones
__device__ __forceinline__ void gather_row(const GatherArgs &g, int64_t row, int tid, int nthreads)
{
    const int64_t b = row < g.batch ? row : row - g.batch;
    const int64_t at = (g.base ? *g.base : 0) + g.base_add + b;
    if (g.n_pairs >= 0 && at >= g.n_pairs) return;
    const int64_t pair = g.pair_idx[at];
    const int64_t m = pair / g.n, s = pair - m * g.n;
    const float *src = g.feats + (row < g.batch ? 0 : (m + 1) * g.view_stride) + s * g.f;
    float *dst = g.y + row * g.f;
    if ((g.f & 3) == 0) {
        const float4 *src4 = (const float4 *)src;
        float4 *dst4 = (float4 *)dst;
        // the row is a random 4^k * 4 B read from HBM: put four independent 16-byte loads per thread in flight before any arithmetic
        const int64_t n4 = g.f / 4;
        for (int64_t i0 = tid; i0 < n4; i0 += 4 * (int64_t)nthreads) {
            float4 vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + (int64_t)u * nthreads;
                if (i < n4) vv[u] = src4[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + (int64_t)u * nthreads;
                if (i >= n4) continue;
                const float4 v = vv[u];
                const int64_t c = i * 4;
                float4 o;
                if (g.inv_scale != nullptr) {
                    o.x = std_f32_rcp(v.x, g.mean[c + 0], g.scale[c + 0], g.inv_scale[c + 0]);
                    o.y = std_f32_rcp(v.y, g.mean[c + 1], g.scale[c + 1], g.inv_scale[c + 1]);
                    o.z = std_f32_rcp(v.z, g.mean[c + 2], g.scale[c + 2], g.inv_scale[c + 2]);
                    o.w = std_f32_rcp(v.w, g.mean[c + 3], g.scale[c + 3], g.inv_scale[c + 3]);
                } else {
                    o.x = std_f32(v.x, g.mean[c + 0], g.scale[c + 0]);
                    o.y = std_f32(v.y, g.mean[c + 1], g.scale[c + 1]);
                    o.z = std_f32(v.z, g.mean[c + 2], g.scale[c + 2]);
                    o.w = std_f32(v.w, g.mean[c + 3], g.scale[c + 3]);
                }
                dst4[i] = o;
            }
        }
    } else {
        for (int64_t i = tid; i < g.f; i += nthreads) dst[i] = std_f32(src[i], g.mean[i], g.scale[i]);
    }
}

// Tiled form of the same copy for 4 | f (used by every launcher): one 256-thread workgroup owns GATHER_ROWS output rows x 1024
// columns.  A thread reads its 4 columns' (mean, scale, 1/scale) once for all the rows -- per element those tables are 6x the
// feature bytes -- and has GATHER_ROWS independent 16-byte row reads in flight.
constexpr int GATHER_ROWS = 8;

__host__ __device__ inline int64_t gather_blocks(int64_t f, int64_t batch)
{
    if (f & 3) return 2 * batch;
    return ((2 * batch + GATHER_ROWS - 1) / GATHER_ROWS) * ((f / 4 + 255) / 256);
}

// R output rows x 1024 columns per 256-thread tile (R independent 16-byte row reads in flight per thread)
template <int R>
__host__ __device__ inline int64_t gather_tiles(int64_t f, int64_t batch)
{
    return ((2 * batch + R - 1) / R) * ((f / 4 + 255) / 256);
}

template <int R>
__device__ __forceinline__ void gather_tile(const GatherArgs &g, int64_t blk, int tid)
{
    const int64_t n4 = g.f / 4, slices = (n4 + 255) / 256;
    const int64_t rg = blk / slices, i = (blk - rg * slices) * 256 + tid;
    if (i >= n4) return;
    const int64_t base = (g.base ? *g.base : 0) + g.base_add;
    const float4 *src4[R];
    float4 vv[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int64_t row = rg * R + u;
        const int64_t b = row < g.batch ? row : row - g.batch;
        const int64_t at = base + b;
        src4[u] = nullptr;
        if (row < 2 * g.batch && (g.n_pairs < 0 || at < g.n_pairs)) {
            const int64_t pair = g.pair_idx[at];
            const int64_t m = pair / g.n, s = pair - m * g.n;
            src4[u] = (const float4 *)(g.feats + (row < g.batch ? 0 : (m + 1) * g.view_stride) + s * g.f);
        }
    }
#pragma unroll
    for (int u = 0; u < R; ++u)
        if (src4[u] != nullptr) vv[u] = src4[u][i];
    const int64_t c = i * 4;
    double mu[4], sc[4], rc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { mu[e] = g.mean[c + e]; sc[e] = g.scale[c + e]; rc[e] = g.inv_scale ? g.inv_scale[c + e] : 0.0; }
#pragma unroll
    for (int u = 0; u < R; ++u) {
        if (src4[u] == nullptr) continue;
        const float4 v = vv[u];
        float4 o;
        if (g.inv_scale != nullptr) {
            o.x = std_f32_rcp(v.x, mu[0], sc[0], rc[0]); o.y = std_f32_rcp(v.y, mu[1], sc[1], rc[1]);
            o.z = std_f32_rcp(v.z, mu[2], sc[2], rc[2]); o.w = std_f32_rcp(v.w, mu[3], sc[3], rc[3]);
        } else {
            o.x = std_f32(v.x, mu[0], sc[0]); o.y = std_f32(v.y, mu[1], sc[1]);
            o.z = std_f32(v.z, mu[2], sc[2]); o.w = std_f32(v.w, mu[3], sc[3]);
        }
        if (g.y != nullptr) ((float4 *)(g.y + (rg * R + u) * g.f))[i] = o;      // (NULL with planes: every consumer of the batch reads the planes)
        if (g.yh != nullptr) {
            constexpr float ps = (float)(1 << idl_planes::X_EXP);
            uint2 h, l;
            if (idl_planes::split4(o.x * ps, o.y * ps, o.z * ps, o.w * ps, h, l) && g.over != nullptr) *g.over = 1;
            ((uint2 *)(g.yh + (rg * R + u) * g.f))[i] = h;
            ((uint2 *)(g.yl + (rg * R + u) * g.f))[i] = l;
        }
    }
}

__device__ __forceinline__ void gather_block(const GatherArgs &g, int64_t blk, int tid)
{
    if (g.f & 3) { gather_row(g, blk, tid, 256); return; }
    gather_tile<GATHER_ROWS>(g, blk, tid);
}

}  // namespace idl_dev
