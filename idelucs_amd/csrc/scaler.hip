// scaler.hip -- column statistics, standardisation and training-batch assembly on gfx950.
//
// All three are HBM-streaming kernels over the row-major feature store [rows, f]:
//   col_stats    : StandardScaler.fit as the reference uses it (idelucs/utils.py:357-359 on the
//                  float32 "true" view, :404-405 on float64 un-mutated rows): float64 mean and
//                  population variance per column in ONE pass over the data -- sums of (x - x0) and
//                  (x - x0)^2 with x0 = the column's first row, so the variance formula has nothing
//                  to cancel -- within 1e-15 relative of sklearn's two-pass _incremental_mean_and_var
//                  (exactly 0 -> scale 1 for a constant column), deterministic order.
//   standardise  : StandardScaler.transform (utils.py:361-366): (x - mean) then / scale, each
//                  evaluated in float64 and rounded to the array's type.
//   gather_pairs : AugmentedDataset/DataLoader batch collation (utils.py:370-389, :422-429) on the
//                  de-duplicated feature store, fused with `standardise`, so the reference's
//                  [N*n_mimics, 2, F] copy (utils.py:353) never exists in HBM.
// Threads own columns (lane i -> column i, 16-byte vectors where f % 4 == 0), so every row read
// is a contiguous, coalesced segment.
#include "common.h"
#include "scaler_device.h"

namespace {

constexpr int STAT_THREADS = 256;

__host__ __device__ inline int64_t stat_row_blocks(int64_t n)
{
    // enough row blocks to fill the chip at f = 256..4096 columns, few enough that the final
    // sequential combine stays negligible
    int64_t r = (n + 255) / 256;
    if (r > 256) r = 256;
    if (r < 1) r = 1;
    return r;
}

// one pass: partial sums of t = x - x0 and t^2 over a block of rows, V columns per thread (V = 4: 16-byte row reads)
template <typename T, int V>
__global__ __launch_bounds__(STAT_THREADS) void col_shifted_sums_kernel(const T *x, int64_t n, int64_t f, int64_t rows_per_block,
                                                                        double *partial1, double *partial2)
{
    const int64_t c = ((int64_t)blockIdx.x * STAT_THREADS + threadIdx.x) * V;
    if (c >= f) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > n) r1 = n;
    typedef T vec_t __attribute__((ext_vector_type(V)));
    double sh[V], a1[V], a2[V];
    {
        const vec_t v0 = *(const vec_t *)(x + c);
#pragma unroll
        for (int e = 0; e < V; ++e) { sh[e] = (double)v0[e]; a1[e] = 0.0; a2[e] = 0.0; }
    }
    int64_t r = r0;
    for (; r + 4 <= r1; r += 4) {                             // four independent row reads in flight
        vec_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const vec_t *)(x + (r + u) * f + c);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < V; ++e) { const double t = (double)v[u][e] - sh[e]; a1[e] += t; a2[e] += t * t; }
    }
    for (; r < r1; ++r) {
        const vec_t v = *(const vec_t *)(x + r * f + c);
#pragma unroll
        for (int e = 0; e < V; ++e) { const double t = (double)v[e] - sh[e]; a1[e] += t; a2[e] += t * t; }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
        partial1[(int64_t)blockIdx.y * f + c + e] = a1[e];
        partial2[(int64_t)blockIdx.y * f + c + e] = a2[e];
    }
}

// combine the row-block partials: 64 columns per workgroup, 16 groups of row blocks added through LDS in a fixed order
template <typename T>
__global__ __launch_bounds__(1024) void col_finish_kernel(const T *x, const double *partial1, const double *partial2, int64_t blocks,
                                                          int64_t f, int64_t n, double *mean, double *scale)
{
    __shared__ double s1[16][64], s2[16][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    double a1 = 0.0, a2 = 0.0;
    if (c < f)
        for (int64_t b = grp; b < blocks; b += 16) { a1 += partial1[b * f + c]; a2 += partial2[b * f + c]; }
    s1[grp][lane] = a1; s2[grp][lane] = a2;
    __syncthreads();
    if (grp == 0 && c < f) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) { t1 += s1[g][lane]; t2 += s2[g][lane]; }
        const double dn = (double)n;
        mean[c] = (double)x[c] + t1 / dn;
        double var = (t2 - t1 * t1 / dn) / dn;
        if (var < 0.0) var = 0.0;
        double s = sqrt(var);
        if (s < 10.0 * 2.220446049250313e-16) s = 1.0;       // sklearn _handle_zeros_in_scale
        scale[c] = s;
    }
}

using idl_dev::std_f32;

// float64 in: float64 arithmetic throughout, one final rounding to float32 (models.py:163 .type(dtype))
__device__ __forceinline__ float std_f64(double x, double m, double s) { return (float)((x - m) / s); }

__global__ __launch_bounds__(256) void standardise_f32_vec4(const float4 *x, int64_t total4, int64_t f4, const double *mean,
                                                            const double *scale, float4 *y)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = (i % f4) * 4;
        const float4 v = x[i];
        float4 o;
        o.x = std_f32(v.x, mean[c + 0], scale[c + 0]);
        o.y = std_f32(v.y, mean[c + 1], scale[c + 1]);
        o.z = std_f32(v.z, mean[c + 2], scale[c + 2]);
        o.w = std_f32(v.w, mean[c + 3], scale[c + 3]);
        y[i] = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void standardise_scalar(const T *x, int64_t total, int64_t f, const double *mean, const double *scale,
                                                          float *y)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i % f;
        if constexpr (sizeof(T) == 8) y[i] = std_f64((double)x[i], mean[c], scale[c]);
        else y[i] = std_f32((float)x[i], mean[c], scale[c]);
    }
}

// ---------------------------------------------------------------- the predict inputs from integer counts (round 4)
// SequenceDataset (reference idelucs/utils.py:400-405) standardises the un-mutated float64 frequency rows counts / sum(counts) with their
// own StandardScaler.  Materialising those rows costs 3.3 GB written and read twice at cfg2; the same float64 values are formed on
// the fly from the int32 counts (1.6 GB) and the row totals: x = (double)c / (double)T, the expression the vectoriser's float64 output
// uses.  Here as q = c r, q' = fma(fma(-q, T, c), r, q) with r = RN(1 / T): the correctly rounded quotient (Markstein) whenever T's
// significand is not all ones -- an integer below 2^32 never is.
__device__ __forceinline__ double count_freq(int32_t c, double T, double rT)
{
    const double cd = (double)c, q = cd * rT;
    return fma(fma(-q, T, cd), rT, q);
}

constexpr int CS_MAX_THREADS = 1024, CS_MAX_GROUPS = 4;      // f / 4 column quads over <= 1024 threads x <= 4 groups: f <= 16384 (k <= 7)

// Pass A.  A workgroup owns a block of rows and ALL columns (thread -> 4 consecutive columns per group), so that it can form the row
// totals itself (wave sums through LDS, four rows per barrier).  Per column the shifted sums of col_shifted_sums_kernel<double>, in
// the same row order and with the same row blocks: the statistics are bit for bit those of the float64 route.
template <int GROUPS>         // column groups per thread: 1 up to 4096 columns, 4 at 16384 (a template: the arrays below must stay in registers)
__global__ __launch_bounds__(CS_MAX_THREADS) void counts_stats_kernel(const int32_t *counts, int64_t n, int64_t f, int64_t rows_per_block,
                                                                      int32_t *row_totals, double *x0, double *partial1, double *partial2)
{
    constexpr int groups = GROUPS;
    __shared__ int64_t wsum[2][4][CS_MAX_THREADS / 64];
    const int tid = threadIdx.x, nth = blockDim.x, wv = tid >> 6, nw = (nth + 63) >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > n) r1 = n;
    auto total4 = [&](const int4 (&v)[4][GROUPS], int rows, int par, double (&T)[4]) {      // row totals of up to four rows
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t sacc = 0;
            if (u < rows)
                for (int g = 0; g < groups; ++g) sacc += (int64_t)v[u][g].x + v[u][g].y + v[u][g].z + v[u][g].w;
            int part = (int)sacc;                          // (a row's counts add up to its windows + 4^k < 2^31)
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o, 64);
            if ((tid & 63) == 0) wsum[par][u][wv] = part;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) { int64_t t = 0; for (int w = 0; w < nw; ++w) t += wsum[par][u][w]; T[u] = (double)t; }
    };
    // the shift: row 0 of the matrix (every workgroup forms it for itself; the first one publishes it for the finishing kernel)
    double sh[GROUPS][4], a1[GROUPS][4], a2[GROUPS][4];
    {
        int4 v[4][GROUPS];
#pragma unroll
        for (int g = 0; g < groups; ++g) v[0][g] = *(const int4 *)(counts + 4 * ((int64_t)tid + (int64_t)nth * g));
        double T[4];
        total4(v, 1, 0, T);
        const double rT = 1.0 / T[0];
        for (int g = 0; g < groups; ++g) {
            const int32_t c4[4] = {v[0][g].x, v[0][g].y, v[0][g].z, v[0][g].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sh[g][e] = count_freq(c4[e], T[0], rT); a1[g][e] = 0.0; a2[g][e] = 0.0;
                if (blockIdx.x == 0) x0[4 * ((int64_t)tid + (int64_t)nth * g) + e] = sh[g][e];
            }
        }
    }
    int par = 1;
    for (int64_t r = r0; r < r1; r += 4, par ^= 1) {
        const int rows = (int)(r1 - r < 4 ? r1 - r : 4);
        int4 v[4][GROUPS];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < rows)
#pragma unroll
                for (int g = 0; g < groups; ++g) v[u][g] = *(const int4 *)(counts + (r + u) * f + 4 * ((int64_t)tid + (int64_t)nth * g));
        double T[4];
        total4(v, rows, par, T);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u >= rows) break;
            if (tid == 0) row_totals[r + u] = (int32_t)T[u];
            const double rT = 1.0 / T[u];
            for (int g = 0; g < groups; ++g) {
                const int32_t c4[4] = {v[u][g].x, v[u][g].y, v[u][g].z, v[u][g].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { const double t = count_freq(c4[e], T[u], rT) - sh[g][e]; a1[g][e] += t; a2[g][e] += t * t; }
            }
        }
    }
    for (int g = 0; g < groups; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t c = 4 * ((int64_t)tid + (int64_t)nth * g) + e;
            partial1[(int64_t)blockIdx.x * f + c] = a1[g][e];
            partial2[(int64_t)blockIdx.x * f + c] = a2[g][e];
        }
}

// Pass B: rows of int32 counts -> StandardScaler.transform of their float64 frequencies, rounded once to float32 (std_f64)
__global__ __launch_bounds__(256) void counts_standardise_kernel(const int32_t *counts, const int32_t *row_totals, int64_t total4, int64_t f4,
                                                                 const double *mean, const double *scale, float4 *y)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / f4, c = (i - row * f4) * 4;
        const int4 v = *(const int4 *)(counts + 4 * i);
        const double T = (double)row_totals[row], rT = 1.0 / T;
        float4 o;
        o.x = std_f64(count_freq(v.x, T, rT), mean[c + 0], scale[c + 0]);
        o.y = std_f64(count_freq(v.y, T, rT), mean[c + 1], scale[c + 1]);
        o.z = std_f64(count_freq(v.z, T, rT), mean[c + 2], scale[c + 2]);
        o.w = std_f64(count_freq(v.w, T, rT), mean[c + 3], scale[c + 3]);
        y[i] = o;
    }
}

__global__ __launch_bounds__(256) void gather_pairs_kernel(idl_dev::GatherArgs g)
{
    idl_dev::gather_block(g, blockIdx.x, threadIdx.x);
}

}  // namespace

extern "C" {

int64_t idl_col_stats_workspace(int64_t n, int64_t f)
{
    if (n < 0 || f < 0) return -1;
    return 2 * stat_row_blocks(n) * f * (int64_t)sizeof(double);
}

int idl_col_stats(const void *x, int is_f64, int64_t n, int64_t f, double *mean, double *scale,
                  void *workspace, void *stream)
{
    IDL_REQUIRE(n >= 1 && f >= 1, "col_stats needs n >= 1 and f >= 1");
    IDL_REQUIRE(x && mean && scale && workspace, "NULL buffer");
    idl::DeviceInfo di;
    int rc = idl::device_info(&di);
    if (rc != IDL_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t blocks = stat_row_blocks(n);
    const int64_t rpb = (n + blocks - 1) / blocks;
    double *p1 = (double *)workspace, *p2 = p1 + blocks * f;
    const bool vec4 = (f & 3) == 0 && (((uintptr_t)x) & (is_f64 ? 31u : 15u)) == 0;
    const int64_t cols_per_block = (int64_t)STAT_THREADS * (vec4 ? 4 : 1);
    const dim3 grid2((unsigned)((f + cols_per_block - 1) / cols_per_block), (unsigned)blocks);
    const dim3 grid1((unsigned)((f + 63) / 64));
    if (is_f64) {
        if (vec4) hipLaunchKernelGGL((col_shifted_sums_kernel<double, 4>), grid2, dim3(STAT_THREADS), 0, st, (const double *)x, n, f, rpb, p1, p2);
        else hipLaunchKernelGGL((col_shifted_sums_kernel<double, 1>), grid2, dim3(STAT_THREADS), 0, st, (const double *)x, n, f, rpb, p1, p2);
        hipLaunchKernelGGL(col_finish_kernel<double>, grid1, dim3(1024), 0, st, (const double *)x, p1, p2, blocks, f, n, mean, scale);
    } else {
        if (vec4) hipLaunchKernelGGL((col_shifted_sums_kernel<float, 4>), grid2, dim3(STAT_THREADS), 0, st, (const float *)x, n, f, rpb, p1, p2);
        else hipLaunchKernelGGL((col_shifted_sums_kernel<float, 1>), grid2, dim3(STAT_THREADS), 0, st, (const float *)x, n, f, rpb, p1, p2);
        hipLaunchKernelGGL(col_finish_kernel<float>, grid1, dim3(1024), 0, st, (const float *)x, p1, p2, blocks, f, n, mean, scale);
    }
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_standardise(const void *x, int is_f64, int64_t n, int64_t f, const double *mean,
                    const double *scale, float *y, void *stream)
{
    IDL_REQUIRE(n >= 0 && f >= 1, "standardise needs n >= 0 and f >= 1");
    if (n == 0) return IDL_OK;
    IDL_REQUIRE(x && mean && scale && y, "NULL buffer");
    idl::DeviceInfo di;
    int rc = idl::device_info(&di);
    if (rc != IDL_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = n * f;
    const int max_blocks = di.cus * 8;
    if (!is_f64 && (f & 3) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15u) == 0) {
        const int64_t total4 = total / 4;
        int64_t g = (total4 + 255) / 256;
        if (g > max_blocks) g = max_blocks;
        hipLaunchKernelGGL(standardise_f32_vec4, dim3((unsigned)g), dim3(256), 0, st, (const float4 *)x, total4, f / 4, mean, scale,
                           (float4 *)y);
    } else {
        int64_t g = (total + 255) / 256;
        if (g > max_blocks) g = max_blocks;
        if (is_f64)
            hipLaunchKernelGGL(standardise_scalar<double>, dim3((unsigned)g), dim3(256), 0, st, (const double *)x, total, f, mean, scale, y);
        else
            hipLaunchKernelGGL(standardise_scalar<float>, dim3((unsigned)g), dim3(256), 0, st, (const float *)x, total, f, mean, scale, y);
    }
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_gather_pairs_at(const float *feats, int64_t n, int64_t f, int64_t view_stride,
                        const int64_t *pair_idx, const int64_t *base, int64_t batch, const double *mean,
                        const double *scale, const double *inv_scale, float *y, void *stream);

int idl_gather_pairs(const float *feats, int64_t n, int64_t f, int64_t view_stride,
                     const int64_t *pair_idx, int64_t batch, const double *mean, const double *scale,
                     float *y, void *stream)
{
    return idl_gather_pairs_at(feats, n, f, view_stride, pair_idx, nullptr, batch, mean, scale, nullptr, y, stream);
}

int idl_gather_pairs_at(const float *feats, int64_t n, int64_t f, int64_t view_stride,
                        const int64_t *pair_idx, const int64_t *base, int64_t batch, const double *mean,
                        const double *scale, const double *inv_scale, float *y, void *stream)
{
    IDL_REQUIRE(n >= 1 && f >= 1 && batch >= 0, "gather_pairs needs n >= 1, f >= 1, batch >= 0");
    if (batch == 0) return IDL_OK;
    IDL_REQUIRE(feats && pair_idx && mean && scale && y, "NULL buffer");
    IDL_REQUIRE((f & 3) != 0 || ((((uintptr_t)feats | (uintptr_t)y) & 15u) == 0 && (view_stride & 3) == 0),
                "feats/y must be 16-byte aligned and view_stride a multiple of 4 when f % 4 == 0");
    idl::DeviceInfo di;
    int rc = idl::device_info(&di);
    if (rc != IDL_OK) return rc;
    idl_dev::GatherArgs g{feats, n, f, view_stride, pair_idx, base, batch, (int64_t)-1, mean, scale, inv_scale, y};
    hipLaunchKernelGGL(gather_pairs_kernel, dim3((unsigned)idl_dev::gather_blocks(f, batch)), dim3(256), 0, (hipStream_t)stream, g);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

/* StandardScaler statistics of the float64 frequency rows counts / sum(counts) straight from int32 counts [n, f] (f % 4 == 0, f <=
 * 16384): mean / scale as idl_col_stats gives them on the materialised float64 rows, bit for bit; row_totals[n] receives every
 * row's sum.  workspace: idl_counts_stats_workspace(n, f) bytes. */
int64_t idl_counts_stats_workspace(int64_t n, int64_t f)
{
    if (n < 0 || f < 0) return -1;
    return (2 * stat_row_blocks(n) + 1) * f * (int64_t)sizeof(double);
}

int idl_counts_stats(const int32_t *counts, int64_t n, int64_t f, double *mean, double *scale, int32_t *row_totals, void *workspace,
                     void *stream)
{
    IDL_REQUIRE(n >= 1 && f >= 4 && (f & 3) == 0 && f <= 4 * CS_MAX_THREADS * CS_MAX_GROUPS, "counts_stats needs n >= 1, 4 | f, f <= 16384");
    IDL_REQUIRE(counts && mean && scale && row_totals && workspace && (((uintptr_t)counts) & 15u) == 0, "counts_stats: NULL or misaligned buffer");
    const int64_t q = f / 4;
    int threads = (int)(q < CS_MAX_THREADS ? q : CS_MAX_THREADS);
    threads = (threads + 63) / 64 * 64;
    IDL_REQUIRE(q % threads == 0, "counts_stats: f / 4 must be a multiple of 64 (or of 1024 beyond 4096 columns)");
    const int groups = (int)(q / threads);
    IDL_REQUIRE(groups == 1 || groups == 4, "counts_stats: f / 4 is 64 .. 1024 or 4096");
    hipStream_t st = (hipStream_t)stream;
    const int64_t blocks = stat_row_blocks(n);
    const int64_t rpb = (n + blocks - 1) / blocks;
    double *p1 = (double *)workspace, *p2 = p1 + blocks * f, *x0 = p2 + blocks * f;
    if (groups == 1) hipLaunchKernelGGL(counts_stats_kernel<1>, dim3((unsigned)blocks), dim3((unsigned)threads), 0, st, counts, n, f, rpb, row_totals, x0, p1, p2);
    else hipLaunchKernelGGL(counts_stats_kernel<4>, dim3((unsigned)blocks), dim3((unsigned)threads), 0, st, counts, n, f, rpb, row_totals, x0, p1, p2);
    hipLaunchKernelGGL(col_finish_kernel<double>, dim3((unsigned)((f + 63) / 64)), dim3(1024), 0, st, (const double *)x0, p1, p2, blocks, f, n, mean, scale);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

/* rows [0, n) of `counts` (with their totals) -> float32 StandardScaler.transform of their float64 frequencies (idl_standardise on
 * the materialised float64 rows, bit for bit).  The pointers may address a row shard of a larger matrix. */
int idl_counts_standardise(const int32_t *counts, const int32_t *row_totals, int64_t n, int64_t f, const double *mean, const double *scale,
                           float *y, void *stream)
{
    IDL_REQUIRE(n >= 0 && f >= 4 && (f & 3) == 0, "counts_standardise needs n >= 0 and 4 | f");
    if (n == 0) return IDL_OK;
    IDL_REQUIRE(counts && row_totals && mean && scale && y && ((((uintptr_t)counts) | ((uintptr_t)y)) & 15u) == 0, "counts_standardise: NULL or misaligned buffer");
    idl::DeviceInfo di;
    int rc = idl::device_info(&di);
    if (rc != IDL_OK) return rc;
    const int64_t total4 = n * f / 4;
    int64_t g = (total4 + 255) / 256;
    if (g > (int64_t)di.cus * 16) g = (int64_t)di.cus * 16;
    hipLaunchKernelGGL(counts_standardise_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, counts, row_totals, total4, f / 4, mean, scale,
                       (float4 *)y);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

}  // extern "C"
