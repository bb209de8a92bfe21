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

// ---------------------------------------------------------------- ... and for 48 < C <= 200 (the fine-grained mode): two launches
// One workgroup took 32 us for the 40 000 elements of a C = 200 joint (s_memtime stamps: 8 us for the row marginals, 10 us for the
// elementwise pass, 5 us each for the first read and the last write -- serial chains on the 4 SIMDs of ONE CU).  A single launch
// on several CUs with a last-workgroup-done counter was no faster: the two agent-scope fences it needs cost 4-8 us EACH on this
// GPU (its L2s are per XCD: a release writes the L2 back, an acquire invalidates it).  So the launch boundary is the barrier:
//   iic_core_rows_multi, ceil(C / 8) workgroups: every one keeps the whole joint in (dynamic) LDS -- one coalesced read of the
//     160 KB, rows padded to an odd stride so that the transposed reads of the symmetrisation are conflict-free -- and repeats the
//     cheap parts (total, the C row marginals) for itself, so no workgroup waits for another; the expensive elementwise pass (loss
//     terms + gradient) is done for IIC_RPW rows per workgroup, two waves per row.  The rows go to `grad` unshifted, with two
//     partial sums per workgroup (and the total s) in `part`.
//   iic_core_shift: the gradient needs the global sum gp = sum g p: every workgroup adds the partials up in workgroup order and
//     writes w_iic (g - gp) / s over its slice of P0; workgroup 0 writes the loss.
// L: C * (C | 1) floats of LDS.  part: 2 * gridDim.x + 1 doubles.  1024 threads.
constexpr int IIC_RPW = 8;
constexpr int IIC_U = 10;      // loads in flight per thread (a C = 200 joint is 9.8 float4 per thread)

__device__ __forceinline__ void iic_core_rows_multi(const float *P0, int C, float lamb, float eps, float *grad, double *part, float *L)
{
    constexpr int NT = 1024, NW = NT / 64;
    __shared__ float rs[200], ar[200];
    __shared__ double red[2][NW];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, n = C * C, G = gridDim.x;
    const int CP = C | 1;
    const bool vec = (C & 3) == 0 && (((uintptr_t)P0) & 15u) == 0;
    // (loads issued IIC_U at a time before the first use: a loop of load -> wait -> use pays the memory latency once per element)
    double acc = 0.0;
    if (vec) {
        for (int b4 = 0; b4 < n / 4; b4 += IIC_U * NT) {
            float4 v[IIC_U];
#pragma unroll
            for (int j = 0; j < IIC_U; ++j) { const int i4 = b4 + j * NT + t; v[j] = ((const float4 *)P0)[i4 < n / 4 ? i4 : 0]; }
#pragma unroll
            for (int j = 0; j < IIC_U; ++j) {
                const int i4 = b4 + j * NT + t;
                if (i4 < n / 4) {
                    const int r = (4 * i4) / C, c = 4 * i4 - r * C;
                    float *d = L + r * CP + c;
                    d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
                    acc += ((double)v[j].x + (double)v[j].y) + ((double)v[j].z + (double)v[j].w);
                }
            }
        }
    } else {
        for (int b = 0; b < n; b += IIC_U * NT) {
            float v[IIC_U];
#pragma unroll
            for (int j = 0; j < IIC_U; ++j) { const int i = b + j * NT + t; v[j] = P0[i < n ? i : 0]; }
#pragma unroll
            for (int j = 0; j < IIC_U; ++j) {
                const int i = b + j * NT + t;
                if (i < n) { const int r = i / C; L[r * CP + (i - r * C)] = v[j]; acc += (double)v[j]; }
            }
        }
    }
    acc = idl_dev::wave_sum_d(acc);
    if (lane == 0) red[0][wv] = acc;
    __syncthreads();
    double st = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) st += red[0][i];
    const float s = (float)st;                                // the same value in every workgroup (same data, same order)
    // the marginals of ALL rows (this pass is repeated by every workgroup: a reciprocal instead of 40 000 divisions), one wave per
    // row, four rows in flight per wave (one row at a time is a serial chain of LDS latency, two wave reductions and a log)
    const float half_inv_s = 0.5f / s;
    for (int r0 = 4 * wv; r0 < C; r0 += 4 * NW) {
        float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = lane; c < C; c += 64) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = min(r0 + j, C - 1);
                const float p = (L[r * CP + c] + L[c * CP + r]) * half_inv_s;
                a[j] += p; b[j] += fmaxf(p, eps);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = idl_dev::wave_sum_f(a[j]); b[j] = idl_dev::wave_sum_f(b[j]); }
        if (lane < 4 && r0 + lane < C) {
            const float al = lane == 0 ? a[0] : lane == 1 ? a[1] : lane == 2 ? a[2] : a[3];
            const float bl = lane == 0 ? b[0] : lane == 1 ? b[1] : lane == 2 ? b[2] : b[3];
            const float pi = fmaxf(al, eps);
            rs[r0 + lane] = __logf(pi);                             // log of the clamped marginal
            ar[r0 + lane] = (al < eps) ? 0.f : lamb * bl / pi;      // its share of dL/dP (none through a clamped marginal)
        }
    }
    __syncthreads();
    // loss terms and gradient of this workgroup's rows: two waves per row
    double lacc = 0.0, gacc = 0.0;
    const int r = blockIdx.x * IIC_RPW + (wv >> 1);
    if (r < C && (wv >> 1) < IIC_RPW) {
        const float lpi = rs[r], gri = ar[r];
        for (int c = (wv & 1) * 64 + lane; c < C; c += 128) {
            const float pu = ((L[r * CP + c] + L[c * CP + r]) * 0.5f) / s, p = fmaxf(pu, eps);
            const float lg = __logf(p) - lamb * rs[c] - lamb * lpi;
            lacc += (double)(-p * lg);
            float g = 0.f;
            if (!(pu < eps)) g += -lg - 1.f;
            g += gri;
            g += ar[c];
            grad[r * C + c] = g;
            gacc += (double)g * (double)pu;
        }
    }
    lacc = idl_dev::wave_sum_d(lacc); gacc = idl_dev::wave_sum_d(gacc);
    if (lane == 0) { red[0][wv] = lacc; red[1][wv] = gacc; }       // (red[0] was last read before the barrier above)
    __syncthreads();
    if (t == 0) {
        double l1 = 0.0, g1 = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) { l1 += red[0][i]; g1 += red[1][i]; }
        part[2 * blockIdx.x] = l1; part[2 * blockIdx.x + 1] = g1;
        if (blockIdx.x == 0) part[2 * G] = (double)s;
    }
}

// P0[i] = w_iic (grad[i] - gp) / s for the 4 * blockDim.x elements of this workgroup; n_part = the grid of iic_core_rows_multi
__device__ __forceinline__ void iic_core_shift(float *P0, int C, float w_iic, float *out, const float *grad, const double *part, int n_part)
{
    const int n = C * C, i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    float g[4];
    const bool vec = (n & 3) == 0 && ((((uintptr_t)P0) | ((uintptr_t)grad)) & 15u) == 0;
    if (vec && i0 < n) { const float4 v = *(const float4 *)(grad + i0); g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w; }
    else {
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = grad[min(i0 + j, n - 1)];
    }
    double l1 = 0.0, g1 = 0.0;
    for (int i = 0; i < n_part; ++i) { l1 += part[2 * i]; g1 += part[2 * i + 1]; }
    const float gp = (float)g1, s = (float)part[2 * n_part];
    if (blockIdx.x == 0 && threadIdx.x == 0) out[3] = (float)l1;
    if (vec && i0 < n) *(float4 *)(P0 + i0) = float4{w_iic * (g[0] - gp) / s, w_iic * (g[1] - gp) / s, w_iic * (g[2] - gp) / s, w_iic * (g[3] - gp) / s};
    else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i0 + j < n) P0[i0 + j] = w_iic * (g[j] - gp) / s;
    }
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

