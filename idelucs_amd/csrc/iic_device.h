// iic_device.h -- the IIC core as a device function, shared by train_step.hip (its own launch) and nce_fused.hip (where
// it rides along as one extra workgroup of InfoNCE pass 1: the two loss branches are independent).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wave_ops.h"

namespace idl_dev {
__device__ __forceinline__ float wsum_f(float v) { return wave_sum_f(v); }
__device__ __forceinline__ double wsum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
}  // namespace idl_dev

// ---------------------------------------------------------------- IIC on the C x C joint (one workgroup)
// P0 = z1^T z2 (given).  Writes IIC to out[3] and dP0 = w_iic * dIIC/dP0 into P0 in place.  scratch: C*C floats.
// (The step loss is assembled by rmsprop_kernel, so that this kernel does not depend on the InfoNCE branch.)
// The joint P = (P0 + P0^T) / (2 sum P0) is symmetric bit-for-bit ((a+b) == (b+a)), so its column sums equal its
// row sums and dL/dP is symmetric too: the reference's two marginals (LossFunctions.py:32-33) and its
// symmetrisation backward collapse to one coalesced wave-per-row pass and no transpose.
template <int NT>
__device__ __forceinline__ void iic_core_body(float *P0, int C, float lamb, float eps, float w_iic, float *scratch, float *out)
{
    constexpr int NW = NT / 64;
    __shared__ double red[NW];
    __shared__ float rs[256], ar[256];          // row sums of P and of the clamped P
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, n = C * C;
    auto block_sum = [&](double v) -> double {
        v = idl_dev::wsum_d(v);
        __syncthreads();
        if (lane == 0) red[wv] = v;
        __syncthreads();
        double r = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) r += red[i];
        return r;
    };
    double acc = 0.0;
    for (int i = t; i < n; i += NT) acc += (double)P0[i];
    const float s = (float)block_sum(acc);
    for (int i = t; i < n; i += NT) {
        const int r = i / C, c = i - r * C;
        scratch[i] = ((P0[i] + P0[c * C + r]) * 0.5f) / s;
    }
    __syncthreads();
    for (int r = wv; r < C; r += NW) {          // one wave per row, coalesced
        float a = 0.f, b = 0.f;
        for (int c = lane; c < C; c += 64) { const float p = scratch[r * C + c]; a += p; b += fmaxf(p, eps); }
        a = idl_dev::wsum_f(a); b = idl_dev::wsum_f(b);
        if (lane == 0) { rs[r] = a; ar[r] = b; }
    }
    __syncthreads();
    // loss, G = dL/dP (clamp-by-assignment: no gradient through clamped entries) and sum(G * P) in one pass
    double lacc = 0.0, gacc = 0.0;
    for (int i = t; i < n; i += NT) {
        const int r = i / C, c = i - r * C;
        const float pu = scratch[i], p = fmaxf(pu, eps);
        const float piu = rs[r], pi = fmaxf(piu, eps), pju = rs[c], pj = fmaxf(pju, eps);
        const float lg = __logf(p) - lamb * __logf(pj) - lamb * __logf(pi);
        lacc += (double)(-p * lg);
        float g = 0.f;
        if (!(pu < eps)) g += -lg - 1.f;
        if (!(piu < eps)) g += lamb * ar[r] / pi;
        if (!(pju < eps)) g += lamb * ar[c] / pj;
        P0[i] = g;
        gacc += (double)g * (double)pu;
    }
    const float iic = (float)block_sum(lacc);
    const float gp = (float)block_sum(gacc);
    if (t == 0) out[3] = iic;
    // through P = Ps / sum(Ps): dPs = (G - sum(G P)) / s; G is symmetric, so (dPs + dPs^T)/2 = dPs
    for (int i = t; i < n; i += NT) P0[i] = w_iic * (P0[i] - gp) / s;
}

// ---------------------------------------------------------------- the same core for C <= 48, resident in LDS
// One 256-thread workgroup.  The joint (<= 9 KB) is read from memory once and dP0 written once; everything between -- the
// symmetrised normalised joint, its marginals, the loss and the gradient -- lives in LDS, and the three float64 block
// reductions go through the VALU-only wave all-reduce.  Same arithmetic, element for element, as iic_core_body (the order of
// the float64 additions inside a block sum differs, below float32 resolution of the results).
constexpr int IIC_SMALL_C = 48;

__device__ __forceinline__ void iic_core_small(float *P0, int C, float lamb, float eps, float w_iic, float *out)
{
    constexpr int NT = 256, NW = 4;
    __shared__ float Pl[IIC_SMALL_C * IIC_SMALL_C], Ps[IIC_SMALL_C * IIC_SMALL_C];
    __shared__ float rs[IIC_SMALL_C], ar[IIC_SMALL_C];
    __shared__ double red[2][NW];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, n = C * C;
    double acc = 0.0;
    for (int i = t; i < n; i += NT) { const float p = P0[i]; Pl[i] = p; acc += (double)p; }
    acc = idl_dev::wave_sum_d(acc);
    if (lane == 0) red[0][wv] = acc;
    __syncthreads();
    const float s = (float)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    for (int i = t; i < n; i += NT) {
        const int r = i / C, c = i - r * C;
        Ps[i] = ((Pl[i] + Pl[c * C + r]) * 0.5f) / s;
    }
    __syncthreads();
    for (int r = wv; r < C; r += NW) {          // one wave per row
        const float p = lane < C ? Ps[r * C + lane] : 0.f;
        const float a = idl_dev::wave_sum_f(p), b = idl_dev::wave_sum_f(lane < C ? fmaxf(p, eps) : 0.f);
        if (lane == 0) { rs[r] = a; ar[r] = b; }
    }
    __syncthreads();
    double lacc = 0.0, gacc = 0.0;
    for (int i = t; i < n; i += NT) {
        const int r = i / C, c = i - r * C;
        const float pu = Ps[i], p = fmaxf(pu, eps);
        const float piu = rs[r], pi = fmaxf(piu, eps), pju = rs[c], pj = fmaxf(pju, eps);
        const float lg = __logf(p) - lamb * __logf(pj) - lamb * __logf(pi);
        lacc += (double)(-p * lg);
        float g = 0.f;
        if (!(pu < eps)) g += -lg - 1.f;
        if (!(piu < eps)) g += lamb * ar[r] / pi;
        if (!(pju < eps)) g += lamb * ar[c] / pj;
        Pl[i] = g;
        gacc += (double)g * (double)pu;
    }
    lacc = idl_dev::wave_sum_d(lacc); gacc = idl_dev::wave_sum_d(gacc);
    if (lane == 0) { red[0][wv] = lacc; red[1][wv] = gacc; }       // (red[0] was last read before the two barriers above)
    __syncthreads();
    const float iic = (float)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    const float gp = (float)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    if (t == 0) out[3] = iic;
    for (int i = t; i < n; i += NT) P0[i] = w_iic * (Pl[i] - gp) / s;
}

// ---------------------------------------------------------------- the same core for 48 < C <= 256 (fine-grained mode, C = 200)
// One 512-thread workgroup (2 waves per SIMD: 256 VGPRs each); row r of the joint belongs to two neighbouring lanes (r = t / 2),
// each keeping its <= 128 elements in REGISTERS from the first read to the final write: P0 is read once (plus its transpose, for the symmetrisation) and written
// once, the marginals are pair reductions (one DPP step), and only the C row sums go through LDS.  (iic_core_body walked the
// 40 000 elements of a C = 200 joint four times through global memory behind float64 shuffles: 79 us; this: see DESIGN.md.)
__device__ __forceinline__ void iic_core_rows(float *P0, int C, float lamb, float eps, float w_iic, float *out)
{
    constexpr int NW = 8, EPT = 128, G = 2;
    __shared__ float rs[256], ar[256];
    __shared__ double red[2][NW];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, r = t / G, sub = t % G;
    const bool live = r < C;
    float p[EPT];
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int c = sub + G * j;
        p[j] = 0.f;
        if (live && c < C) { const float a = P0[r * C + c], b = P0[c * C + r]; p[j] = (a + b) * 0.5f; acc += (double)a; }
        if ((j & 15) == 15) __builtin_amdgcn_sched_barrier(0);      // 32 loads in flight at a time, not 256 (register budget)
    }
    acc = idl_dev::wave_sum_d(acc);
    if (lane == 0) red[0][wv] = acc;
    __syncthreads();
    double st = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) st += red[0][i];
    const float s = (float)st;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int c = sub + G * j;
        if (live && c < C) { p[j] = p[j] / s; a += p[j]; b += fmaxf(p[j], eps); }
        if ((j & 15) == 15) __builtin_amdgcn_sched_barrier(0);
    }
    a += idl_dev::dpp_f<idl_dev::DPP_XOR1>(a);
    b += idl_dev::dpp_f<idl_dev::DPP_XOR1>(b);
    if (live && sub == 0) { rs[r] = a; ar[r] = b; }
    __syncthreads();
    double lacc = 0.0, gacc = 0.0;
    if (live) {
        const float piu = rs[r], pi = fmaxf(piu, eps), lpi = __logf(pi), ari = ar[r];
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int c = sub + G * j;
            if (c < C) {
                const float pu = p[j], pp = fmaxf(pu, eps);
                const float pju = rs[c], pj = fmaxf(pju, eps);
                const float lg = __logf(pp) - lamb * __logf(pj) - lamb * lpi;
                lacc += (double)(-pp * lg);
                float g = 0.f;
                if (!(pu < eps)) g += -lg - 1.f;
                if (!(piu < eps)) g += lamb * ari / pi;
                if (!(pju < eps)) g += lamb * ar[c] / pj;
                p[j] = g;
                gacc += (double)g * (double)pu;
            }
            if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
    }
    lacc = idl_dev::wave_sum_d(lacc); gacc = idl_dev::wave_sum_d(gacc);
    if (lane == 0) { red[0][wv] = lacc; red[1][wv] = gacc; }       // (red[0] was last read before the barrier above)
    __syncthreads();
    double l1 = 0.0, g1 = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) { l1 += red[0][i]; g1 += red[1][i]; }
    const float iic = (float)l1, gp = (float)g1;
    if (t == 0) out[3] = iic;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int c = sub + G * j;
        if (live && c < C) P0[r * C + c] = w_iic * (p[j] - gp) / s;
        if ((j & 15) == 15) __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------- ... and for C x C x 4 B <= ~158 KB (C <= 200: the fine-grained mode)
// the whole joint in (dynamic) LDS, updated in place: read from memory once with coalesced 16-byte loads, symmetrised pair by pair
// (the thread of element (r, c), r <= c, writes both (r, c) and (c, r)), row sums one wave per row, gradient in place, written
// back once.  L: C * C floats of LDS.  NT threads.
template <int NT>
__device__ __forceinline__ void iic_core_lds(float *P0, int C, float lamb, float eps, float w_iic, float *out, float *L)
{
    constexpr int NW = NT / 64;
    __shared__ float rs[256], ar[256];
    __shared__ double red[2][NW];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, n = C * C;
    double acc = 0.0;
    if ((n & 3) == 0 && ((uintptr_t)P0 & 15u) == 0) {
        for (int i4 = t; i4 < n / 4; i4 += NT) {
            const float4 v = ((const float4 *)P0)[i4];
            *(float4 *)(L + 4 * i4) = v;
            acc += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        }
    } else {
        for (int i = t; i < n; i += NT) { const float v = P0[i]; L[i] = v; acc += (double)v; }
    }
    acc = idl_dev::wave_sum_d(acc);
    if (lane == 0) red[0][wv] = acc;
    __syncthreads();
    double st = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) st += red[0][i];
    const float s = (float)st;
    // everything below walks the joint one wave per row (no integer divisions); the per-row terms of the loss and of the gradient
    // -- log of the clamped marginal, lamb * ar / marginal -- are computed once per row, not once per element
    for (int r = wv; r < C; r += NW)
        for (int c = r + lane; c < C; c += 64) { const float v = ((L[r * C + c] + L[c * C + r]) * 0.5f) / s; L[r * C + c] = v; L[c * C + r] = v; }
    __syncthreads();
    for (int r = wv; r < C; r += NW) {
        float a = 0.f, b = 0.f;
        for (int c = lane; c < C; c += 64) { const float p = L[r * C + c]; a += p; b += fmaxf(p, eps); }
        a = idl_dev::wave_sum_f(a); b = idl_dev::wave_sum_f(b);
        if (lane == 0) {
            const float pi = fmaxf(a, eps);
            rs[r] = __logf(pi);                               // log of the clamped marginal
            ar[r] = (a < eps) ? 0.f : lamb * b / pi;          // its share of dL/dP (none through a clamped marginal)
        }
    }
    __syncthreads();
    double lacc = 0.0, gacc = 0.0;
    for (int r = wv; r < C; r += NW) {
        const float lpi = rs[r], gri = ar[r];
        for (int c = lane; c < C; c += 64) {
            const float pu = L[r * C + c], p = fmaxf(pu, eps);
            const float lg = __logf(p) - lamb * rs[c] - lamb * lpi;
            lacc += (double)(-p * lg);
            float g = 0.f;
            if (!(pu < eps)) g += -lg - 1.f;
            g += gri;
            g += ar[c];
            L[r * C + c] = g;
            gacc += (double)g * (double)pu;
        }
    }
    lacc = idl_dev::wave_sum_d(lacc); gacc = idl_dev::wave_sum_d(gacc);
    if (lane == 0) { red[0][wv] = lacc; red[1][wv] = gacc; }       // (red[0] was last read two barriers ago)
    __syncthreads();
    double l1 = 0.0, g1 = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) { l1 += red[0][i]; g1 += red[1][i]; }
    const float gp = (float)g1;
    if (t == 0) out[3] = (float)l1;
    for (int i = t; i < n; i += NT) P0[i] = w_iic * (L[i] - gp) / s;      // (each thread re-reads only what it wrote)
}

// ---------------------------------------------------------------- the C <= 48 core, from a read-only joint into LDS
// For the fused InfoNCE-pass-2 + middle-backward kernel: every one of its workgroups needs w_iic * dIIC/dP0, and recomputing the
// 400..2304-element core per workgroup is cheaper than a launch boundary.  P0: the joint in memory (not modified); Pl (C * C floats
// of LDS) receives the gradient; Ps: C * C floats of LDS scratch; out (may be NULL): out[3] = IIC.  Ends with a barrier.
template <int NT>
__device__ __forceinline__ void iic_core_to_lds(const float *P0, int C, float lamb, float eps, float w_iic, float *out, float *Pl, float *Ps)
{
    constexpr int NW = NT / 64;
    __shared__ float rs[IIC_SMALL_C], ar[IIC_SMALL_C];
    __shared__ double red[2][NW];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, n = C * C;
    double acc = 0.0;
    for (int i = t; i < n; i += NT) { const float p = P0[i]; Pl[i] = p; acc += (double)p; }
    acc = idl_dev::wave_sum_d(acc);
    if (lane == 0) red[0][wv] = acc;
    __syncthreads();
    double st = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) st += red[0][i];
    const float s = (float)st;
    for (int i = t; i < n; i += NT) {
        const int r = i / C, c = i - r * C;
        Ps[i] = ((Pl[i] + Pl[c * C + r]) * 0.5f) / s;
    }
    __syncthreads();
    for (int r = wv; r < C; r += NW) {          // one wave per row
        const float p = lane < C ? Ps[r * C + lane] : 0.f;
        const float a = idl_dev::wave_sum_f(p), b = idl_dev::wave_sum_f(lane < C ? fmaxf(p, eps) : 0.f);
        if (lane == 0) { rs[r] = a; ar[r] = b; }
    }
    __syncthreads();
    double lacc = 0.0, gacc = 0.0;
    for (int i = t; i < n; i += NT) {
        const int r = i / C, c = i - r * C;
        const float pu = Ps[i], p = fmaxf(pu, eps);
        const float piu = rs[r], pi = fmaxf(piu, eps), pju = rs[c], pj = fmaxf(pju, eps);
        const float lg = __logf(p) - lamb * __logf(pj) - lamb * __logf(pi);
        lacc += (double)(-p * lg);
        float g = 0.f;
        if (!(pu < eps)) g += -lg - 1.f;
        if (!(piu < eps)) g += lamb * ar[r] / pi;
        if (!(pju < eps)) g += lamb * ar[c] / pj;
        Pl[i] = g;
        gacc += (double)g * (double)pu;
    }
    lacc = idl_dev::wave_sum_d(lacc); gacc = idl_dev::wave_sum_d(gacc);
    if (lane == 0) { red[0][wv] = lacc; red[1][wv] = gacc; }
    __syncthreads();
    double l1 = 0.0, g1 = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) { l1 += red[0][i]; g1 += red[1][i]; }
    const float gp = (float)g1;
    if (out != nullptr && t == 0) out[3] = (float)l1;
    for (int i = t; i < n; i += NT) Pl[i] = w_iic * (Pl[i] - gp) / s;
    __syncthreads();
}

