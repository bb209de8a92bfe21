// scaler.hip -- column statistics, standardisation and training-batch assembly on gfx950.
//
// All three are HBM-streaming kernels over the row-major feature store [rows, f]:
//   col_stats    : StandardScaler.fit as the reference uses it (idelucs/utils.py:357-359 on the
//                  float32 "true" view, :404-405 on float64 un-mutated rows): float64 mean and
//                  population variance per column, two passes (sum; centred sum + centred sum of
//                  squares) exactly like sklearn's _incremental_mean_and_var, deterministic order.
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

template <typename T>
__global__ __launch_bounds__(STAT_THREADS) void col_sum_kernel(const T *x, int64_t n, int64_t f, int64_t rows_per_block,
                                                               double *partial)
{
    const int64_t c = (int64_t)blockIdx.x * STAT_THREADS + threadIdx.x;
    if (c >= f) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > n) r1 = n;
    double acc = 0.0;
    for (int64_t r = r0; r < r1; ++r) acc += (double)x[r * f + c];
    partial[(int64_t)blockIdx.y * f + c] = acc;
}

__global__ __launch_bounds__(STAT_THREADS) void col_mean_kernel(const double *partial, int64_t blocks, int64_t f, int64_t n,
                                                                double *mean)
{
    const int64_t c = (int64_t)blockIdx.x * STAT_THREADS + threadIdx.x;
    if (c >= f) return;
    double acc = 0.0;
    for (int64_t b = 0; b < blocks; ++b) acc += partial[b * f + c];
    mean[c] = acc / (double)n;
}

template <typename T>
__global__ __launch_bounds__(STAT_THREADS) void col_centred_kernel(const T *x, int64_t n, int64_t f, int64_t rows_per_block,
                                                                   const double *mean, double *partial1, double *partial2)
{
    const int64_t c = (int64_t)blockIdx.x * STAT_THREADS + threadIdx.x;
    if (c >= f) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > n) r1 = n;
    const double m = mean[c];
    double a1 = 0.0, a2 = 0.0;
    for (int64_t r = r0; r < r1; ++r) {
        const double t = (double)x[r * f + c] - m;
        a1 += t;
        a2 += t * t;
    }
    partial1[(int64_t)blockIdx.y * f + c] = a1;
    partial2[(int64_t)blockIdx.y * f + c] = a2;
}

__global__ __launch_bounds__(STAT_THREADS) void col_scale_kernel(const double *partial1, const double *partial2, int64_t blocks,
                                                                 int64_t f, int64_t n, double *scale)
{
    const int64_t c = (int64_t)blockIdx.x * STAT_THREADS + threadIdx.x;
    if (c >= f) return;
    double corr = 0.0, ss = 0.0;
    for (int64_t b = 0; b < blocks; ++b) {
        corr += partial1[b * f + c];
        ss += partial2[b * f + c];
    }
    const double var = (ss - corr * corr / (double)n) / (double)n;
    double s = sqrt(var);
    if (s < 10.0 * 2.220446049250313e-16) s = 1.0;  // sklearn _handle_zeros_in_scale
    scale[c] = s;
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
    const dim3 grid2((unsigned)((f + STAT_THREADS - 1) / STAT_THREADS), (unsigned)blocks);
    const dim3 grid1((unsigned)((f + STAT_THREADS - 1) / STAT_THREADS));
    if (is_f64) hipLaunchKernelGGL(col_sum_kernel<double>, grid2, dim3(STAT_THREADS), 0, st, (const double *)x, n, f, rpb, p1);
    else hipLaunchKernelGGL(col_sum_kernel<float>, grid2, dim3(STAT_THREADS), 0, st, (const float *)x, n, f, rpb, p1);
    hipLaunchKernelGGL(col_mean_kernel, grid1, dim3(STAT_THREADS), 0, st, p1, blocks, f, n, mean);
    if (is_f64)
        hipLaunchKernelGGL(col_centred_kernel<double>, grid2, dim3(STAT_THREADS), 0, st, (const double *)x, n, f, rpb, mean, p1, p2);
    else
        hipLaunchKernelGGL(col_centred_kernel<float>, grid2, dim3(STAT_THREADS), 0, st, (const float *)x, n, f, rpb, mean, p1, p2);
    hipLaunchKernelGGL(col_scale_kernel, grid1, dim3(STAT_THREADS), 0, st, p1, p2, blocks, f, n, scale);
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

}  // extern "C"
